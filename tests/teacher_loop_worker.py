"""Worker of tests/test_models_gpu.py::test_heun_teacher_loop_at_real_size_graph_equals_eager: BASELINE configs[2] at its
real size (559 M-parameter light teacher U-Net, B = 8 prompts = CFG batch 16, L = 32 text tokens, latent 8 x 256 x 16)
in a fresh process (its own handles, arenas and caches).

    python tests/teacher_loop_worker.py <out.pt> <steps> [notextcache]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import cases  # noqa: E402
from consistencytta_amd import scheduler, spec  # noqa: E402
from consistencytta_amd.models import AudioLCM  # noqa: E402


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    no_cache = len(sys.argv) > 3 and sys.argv[3] == "notextcache"
    dev = "cuda:0"
    cfg = spec.LIGHT_UNET_CONFIG
    B, L = 8, 32
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tango_diffusion_light.json", unet_config=cfg, snr_gamma=5.0, use_edm=True,
                 teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse",
                 target_ema_decay=0.95, ema_decay=0.999)
    m.teacher_unet.load_state_dict(cases.unet_weights(cfg, False, 0))
    m.student_ema_unet.load_state_dict(cases.unet_weights(cfg, True, 3))
    m.to(dev).eval()
    if no_cache:      # every query projects the text states' K / V itself (reuse_text=False semantics)
        m.teacher_unet._text_unchanged = lambda *a, **k: False
    P = {k: v.to(dev) for k, v in cases.prompt_states(cfg, B, L, "teacher_full").items()}
    sched = scheduler.HeunDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-2-1", subfolder="scheduler")
    noise = (cases.t(spec.det_uniform("teacher_full.noise", (B, 8, 256, 16), 16)) * np.float32(np.sqrt(3.0))).to(dev)
    kw = dict(guidance_scale_input=3.0, guidance_scale_post=1.0, num_steps=1, use_edm=True, use_ema=True,
              query_teacher=True, return_all=True, noise=noise, num_teacher_steps=steps)
    stu, eager, _, _ = m.inference(P, sched, **kw)
    _, graphed, _, _ = m.inference(P, sched, graph_teacher=True, **kw)
    _, graphed2, _, _ = m.inference(P, sched, graph_teacher=True, **kw)      # a fresh capture on a warm handle
    torch.cuda.synchronize()
    torch.save({"eager": eager.cpu(), "graphed": graphed.cpu(), "graphed2": graphed2.cpu(), "student": stu.cpu()}, out)


if __name__ == "__main__":
    main()
