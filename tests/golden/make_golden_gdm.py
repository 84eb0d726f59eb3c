"""Fixtures for the stage-1 guided-distillation row, produced by the REFERENCE's own
`diffusers.schedulers.DDPMScheduler / DDIMScheduler` (the batched-timestep versions vendored in /root/reference) and
`models.AudioGDM` (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gdm.py

The two scheduler files are loaded from the FULL vendored diffusers tree into the package context of the trimmed
easy_inference copy (same `ConfigMixin` / `SchedulerMixin` / `BaseOutput` code; the full tree does not import offline).
Scheduler config = the public SD-2.1 scheduler_config.json the reference fetches from the hub
(`scheduler_name="stabilityai/stable-diffusion-2-1"`): scaled_linear betas 0.00085..0.012, v_prediction, clip_sample
false, set_alpha_to_one false."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
import ref_import  # noqa: E402
import make_golden_distill as mgd  # noqa: E402
from consistencytta_amd import spec  # noqa: E402

SD21 = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
            prediction_type="v_prediction", clip_sample=False)


def load_reference_schedulers():
    ref_import.load()
    import diffusers
    import diffusers.utils.configuration_utils as cu
    import diffusers.utils.outputs as outs
    import diffusers.utils.scheduling_utils as su
    import diffusers.utils.torch_utils as tu
    sys.modules["diffusers.configuration_utils"] = cu
    U = sys.modules["diffusers.utils"]
    U.BaseOutput, U.randn_tensor = outs.BaseOutput, tu.randn_tensor
    pkg = types.ModuleType("diffusers.schedulers")
    pkg.__path__ = [os.path.join(ref_import.REF_ROOT, "diffusers", "schedulers")]
    sys.modules["diffusers.schedulers"] = pkg
    sys.modules["diffusers.schedulers.scheduling_utils"] = su
    out = {}
    for name in ("scheduling_ddpm", "scheduling_ddim"):
        spec_ = importlib.util.spec_from_file_location(
            "diffusers.schedulers." + name, os.path.join(pkg.__path__[0], name + ".py"))
        mod = importlib.util.module_from_spec(spec_)
        sys.modules[spec_.name] = mod
        spec_.loader.exec_module(mod)
        out[name] = mod
    return out["scheduling_ddpm"].DDPMScheduler, out["scheduling_ddim"].DDIMScheduler, diffusers


def main():
    DDPM, DDIM, D = load_reference_schedulers()
    out = {}
    B = 3
    x = cases.t(spec.det_uniform("gdm.x", (B, 8, 16, 4), 1)) * 2
    noise = cases.t(spec.det_uniform("gdm.n", (B, 8, 16, 4), 2))
    v = cases.t(spec.det_uniform("gdm.v", (B, 8, 16, 4), 3))
    with torch.no_grad():
        ddpm = DDPM(**SD21)
        t_train = torch.tensor([999, 400, 0])
        out["ddpm_timesteps_head"] = ddpm.timesteps[:5].numpy()
        out["ddpm_add_noise"] = ddpm.add_noise(x, noise, t_train).numpy()
        out["alphas_cumprod"] = ddpm.alphas_cumprod.numpy()
        ddim = DDIM(set_alpha_to_one=False, **SD21)
        for n in (5, 50):
            ddim.set_timesteps(n)
            out["ddim_timesteps_%d" % n] = ddim.timesteps.numpy()
        ddim.set_timesteps(5)
        t_inf = ddim.timesteps[torch.tensor([0, 2, 4])]           # 800, 400, 0 -> the last one steps to "prev < 0"
        out["ddim_t"] = t_inf.numpy()
        out["ddim_step"] = ddim.step(v, t_inf, x).prev_sample.numpy()
        out["ddim_step_scalar_t"] = ddim.step(v, int(ddim.timesteps[1]), x).prev_sample.numpy()
        out["ddim_add_noise"] = ddim.add_noise(x, noise, t_train).numpy()

    # ---- AudioGDM.forward (training loss) and inference with the reference's own class
    ns, _, _ = mgd.load_reference_audiolcm()          # stubs + models package importable
    D.DDPMScheduler, D.DDIMScheduler = DDPM, DDIM
    DDPM.from_pretrained = classmethod(lambda cls, *a, **k: DDPM(**SD21))
    sys.modules.pop("models.audio_guided_model", None)
    from models.audio_guided_model import AudioGDM
    import json
    import tempfile
    cfg = cases.TINY_UNET
    full = dict(json.load(open(ns.light_config_path)))
    full.update(cfg)
    tmp = os.path.join(tempfile.mkdtemp(), "tiny_light.json")
    json.dump(full, open(tmp, "w"))
    torch.manual_seed(0)
    model = AudioGDM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                     unet_model_config_path=tmp, snr_gamma=5.0, teacher_guidance_scale=-1, ema_decay=0.999)
    model.teacher_unet.load_state_dict(cases.unet_weights(cfg, False, 0))
    model.student_unet.load_state_dict(cases.unet_weights(cfg, True, 1))
    model.student_ema_unet.load_state_dict(cases.unet_weights(cfg, True, 3))
    Bm, H, W, L = 3, 32, 8, 6
    P = cases.prompt_states(cfg, Bm, L, "distill")
    model.get_prompt_embeds = lambda prompt, use_cf, num_samples_per_prompt=1: (
        P["embeds_cf"], P["mask_cf"], P["embeds"], P["mask"])
    model.encode_text_classifier_free = lambda prompt, n: (P["embeds_cf"], P["mask_cf"], P["embeds"], P["mask"])
    z0 = cases.t(spec.det_uniform("distill.z0", (Bm, 8, H, W), 14)) * 0.9
    rec = {}
    o_randint, o_randn_like, o_rand = torch.randint, torch.randn_like, torch.rand

    def randint(*a, **k):
        vv = o_randint(*a, **k)
        rec.setdefault("randint", vv.clone())
        return vv

    def randn_like(xx, *a, **k):
        vv = o_randn_like(xx, *a, **k)
        rec.setdefault("randn_like", vv.clone())
        return vv

    def rand(*a, **k):
        vv = o_rand(*a, **k)
        rec.setdefault("rand", vv.clone())
        return vv

    torch.manual_seed(4321)
    torch.randint, torch.randn_like, torch.rand = randint, randn_like, rand
    try:
        model.train()
        model.teacher_unet.eval()
        model.student_ema_unet.eval()
        with torch.no_grad():
            loss = model(z0, ["a"] * Bm)
    finally:
        torch.randint, torch.randn_like, torch.rand = o_randint, o_randn_like, o_rand
    out["gdm_loss"] = float(loss)
    out["gdm_time_inds"] = rec["randint"].numpy()
    out["gdm_noise"] = rec["randn_like"].numpy()
    out["gdm_guidance"] = rec["rand"].numpy() * 6

    # inference: 4 DDIM steps, student (EMA) and teacher side by side, fixed initial noise
    import models.audio_guided_model as AGM
    lat = cases.t(spec.det_uniform("gdm.inf_noise", (Bm, 8, 256, 16), 16)) * np.float32(np.sqrt(3.0))
    AGM.randn_tensor = lambda shape, generator=None, device=None, dtype=None: lat.clone()
    model.eval()
    sched = DDIM(set_alpha_to_one=False, **SD21)
    with torch.no_grad():
        z = model.inference(["a"] * Bm, sched, guidance_scale_input=3.0, guidance_scale_post=1.0, num_steps=4,
                            use_ema=True)
    out["gdm_inference_4steps"] = z.numpy()
    path = os.path.join(HERE, "gdm_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    print({k: (vv if np.ndim(vv) == 0 or np.size(vv) < 8 else np.shape(vv)) for k, vv in out.items()})


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    main()
