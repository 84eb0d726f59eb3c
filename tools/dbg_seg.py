"""Debug aid: segmented step graph vs eager on the tiny model, per-parameter report."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from consistencytta_amd import spec
from consistencytta_amd.models import AudioLCM
DEV = "cuda:0"

def lcm():
    cfg = cases.TINY_UNET
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tiny_light.json", unet_config=cfg, snr_gamma=5.0, use_edm=True,
                 teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse", target_ema_decay=0.95, ema_decay=0.999)
    m.teacher_unet.load_state_dict(cases.unet_weights(cfg, False, 0))
    m.student_unet.load_state_dict(cases.unet_weights(cfg, True, 1))
    m.student_target_unet.load_state_dict(cases.unet_weights(cfg, True, 2))
    m.student_ema_unet.load_state_dict(cases.unet_weights(cfg, True, 3))
    m.to(DEV)
    P = {k: v.to(DEV) for k, v in cases.prompt_states(cfg, 3, 6, "distill").items()}
    z0 = (cases.t(spec.det_uniform("distill.z0", (3, 8, 32, 8), 14)) * 0.9).to(DEV)
    return m, P, z0

prior = sys.argv[1] if len(sys.argv) > 1 else "none"
gen = torch.Generator().manual_seed(11)
kw = dict(time_inds=torch.randint(0, 17, (3,), generator=gen) * 2, gaussian_noise=torch.randn(3, 8, 32, 8, generator=gen).to(DEV),
          guidance_scale=torch.rand(3, generator=gen) * 6)
m_e, P, z0 = lcm(); m_e.train(); o_e = m_e.prepare_training(lr=1e-4, weight_decay=1e-4, broadcast=False)
m_s, _, _ = lcm(); m_s.train(); o_s = m_s.prepare_training(lr=1e-4, weight_decay=1e-4, broadcast=False)
if prior == "eager":
    m_s.train_step(z0, P, o_s, None, **kw)
    for n in ("student_unet", "student_target_unet", "student_ema_unet"):
        getattr(m_s, n)._flat.copy_(getattr(m_e, n)._flat); getattr(m_s, n).mark_weights_changed()
    o_s.zero_grad()
if prior == "mono":
    m_x, _, _ = lcm(); m_x.train(); o_x = m_x.prepare_training(lr=1e-4, weight_decay=1e-4, broadcast=False)
    mono = m_x.capture_train_graph(o_x, z0, P, segmented=False, **kw)
seg = m_s.capture_train_graph(o_s, z0, P, segmented=True, **kw)
with torch.no_grad():
    loss, pred, target, sig, gamma = m_e._forward_impl(z0, None, P, False, True, kw["time_inds"], kw["gaussian_noise"], kw["guidance_scale"], True)
    m_e._student_backward(pred, target, sig, gamma, 1.0, None)
for rep in range(2):
    o_s.zero_grad()
    seg._refresh(z0, kw["time_inds"], kw["gaussian_noise"], kw["guidance_scale"])
    seg.replay()
    torch.cuda.synchronize()
    bad = 0
    for (k, p), (_, q) in zip(m_s.student_unet.named_parameters(), m_e.student_unet.named_parameters()):
        if p.grad is None: continue
        fin = bool(torch.isfinite(p.grad).all())
        rel = float((p.grad - q.grad).norm() / (q.grad.norm() + 1e-30))
        if not fin or rel > 1e-6:
            bad += 1
            if bad <= 12: print("  [%s rep %d] %-60s finite=%s rel=%.3e" % (prior, rep, k, fin, rel))
    print("[%s] rep %d: loss %.9g vs %.9g, %d bad tensors" % (prior, rep, float(seg.loss.item()), float(loss), bad))
