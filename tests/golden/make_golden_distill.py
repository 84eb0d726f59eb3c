"""Fixtures for the distillation / task-level rows, produced by the
REFERENCE's own `models.AudioLCM` (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_distill.py

Import recipe = SURVEY.md §8(c): trimmed diffusers first, then /root/reference with the missing
third-party modules stubbed, a fake FLAN-T5 (seeded states), the SD-2.1 scheduler config in
place of the hub download, and a no-op CLAPLoss.  The random draws the reference makes inside
`forward` (timestep indices, noise, guidance scales) are RECORDED and stored in the fixture so
that other implementations can replay them.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
import ref_import  # noqa: E402
from consistencytta_amd import spec  # noqa: E402


def load_reference_audiolcm():
    ns = ref_import.load()
    import diffusers as D
    import diffusers.utils as DU
    from diffusers.utils import torch_utils
    D.UNet2DConditionModel = ns.UNet2DConditionModel
    D.UNet2DConditionGuidedModel = ns.UNet2DConditionGuidedModel
    D.HeunDiscreteScheduler = ns.HeunDiscreteScheduler
    DU.randn_tensor = torch_utils.randn_tensor

    class _NoSched:
        @classmethod
        def from_pretrained(cls, *a, **k):
            raise RuntimeError("DDPM/DDIM schedulers are not part of the Heun/EDM path")
    D.DDPMScheduler = D.DDIMScheduler = _NoSched
    ns.HeunDiscreteScheduler.from_pretrained = classmethod(lambda cls, *a, **k: ref_import.make_heun(ns))

    ta = sys.modules.get("torchaudio") or types.ModuleType("torchaudio")
    taf = types.ModuleType("torchaudio.functional")
    taf.resample = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no torchaudio"))
    ta.functional = taf
    sys.modules["torchaudio"], sys.modules["torchaudio.functional"] = ta, taf
    lc = sys.modules.get("laion_clap") or types.ModuleType("laion_clap")
    lc.CLAP_Module = object
    sys.modules["laion_clap"] = lc
    for name in ("wandb", "soundfile", "librosa", "resampy"):
        sys.modules.setdefault(name, types.ModuleType(name))

    import accelerate
    accelerate.PartialState(cpu=True)
    import transformers

    class FakeTok:
        model_max_length = 512

        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()

    class FakeT5(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dummy = torch.nn.Parameter(torch.zeros(1))

        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()

        @property
        def device(self):
            return self.dummy.device

    # setattr (not getattr): transformers resolves these lazily and T5's import chain needs soxr here
    for name, obj in (("AutoTokenizer", FakeTok), ("T5EncoderModel", FakeT5), ("CLIPTokenizer", FakeTok),
                      ("CLIPTextModel", FakeT5), ("AutoModel", FakeT5)):
        setattr(transformers, name, obj)

    sys.path.insert(1, ref_import.REF_ROOT)
    import tools.losses as RL

    class NoClap(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    RL.CLAPLoss = NoClap
    from models.audio_consistency_model import AudioLCM
    import tools.train_utils as TU
    return ns, AudioLCM, TU


prompt_states = cases.prompt_states


def main():
    ns, AudioLCM, TU = load_reference_audiolcm()
    cfg = cases.TINY_UNET
    full = dict(json.load(open(ns.light_config_path)))
    full.update(cfg)
    tmp = os.path.join(tempfile.mkdtemp(), "tiny_light.json")   # 'light' in the path, like the real config
    json.dump(full, open(tmp, "w"))
    torch.manual_seed(0)
    model = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                     unet_model_config_path=tmp, snr_gamma=5.0, use_edm=True, teacher_guidance_scale=-1,
                     num_diffusion_steps=18, vae=torch.nn.Identity(), loss_type="mse", target_ema_decay=0.95,
                     ema_decay=0.999)
    model.teacher_unet.load_state_dict(cases.unet_weights(cfg, False, 0))
    model.student_unet.load_state_dict(cases.unet_weights(cfg, True, 1))
    model.student_target_unet.load_state_dict(cases.unet_weights(cfg, True, 2))
    model.student_ema_unet.load_state_dict(cases.unet_weights(cfg, True, 3))
    B, H, W, L = 3, 32, 8, 6
    P = prompt_states(cfg, B, L, "distill")
    model.get_prompt_embeds = lambda prompt, use_cf, num_samples_per_prompt=1: (
        P["embeds_cf"], P["mask_cf"], P["embeds"], P["mask"])
    model.encode_text_classifier_free = lambda prompt, n: (P["embeds_cf"], P["mask_cf"], P["embeds"], P["mask"])
    z0 = cases.t(spec.det_uniform("distill.z0", (B, 8, H, W), 14)) * 0.9
    out = {}

    # ---- record the reference's internal random draws
    rec = {}
    o_randint, o_randn_like, o_rand = torch.randint, torch.randn_like, torch.rand

    def randint(*a, **k):
        v = o_randint(*a, **k)
        rec.setdefault("randint", v.clone())
        return v

    def randn_like(x, *a, **k):
        v = o_randn_like(x, *a, **k)
        rec.setdefault("randn_like", v.clone())
        return v

    def rand(*a, **k):
        v = o_rand(*a, **k)
        rec.setdefault("rand", v.clone())
        return v

    model.eval()
    torch.manual_seed(1234)
    torch.randint, torch.randn_like, torch.rand = randint, randn_like, rand
    try:
        model.train()
        with torch.no_grad():
            loss = model(z0, None, ["a"] * B)
    finally:
        torch.randint, torch.randn_like, torch.rand = o_randint, o_randn_like, o_rand
    out["train_loss"] = float(loss)
    out["time_inds"] = rec["randint"].numpy()          # already multiplied by `order` inside? no: raw draw
    out["noise"] = rec["randn_like"].numpy()
    out["guidance"] = rec["rand"].numpy() * 6          # guidance_scale = rand * max_rand_guidance_scale

    # ---- validation mode (4 losses), fixed timestep index, run_teacher to t=0
    rec.clear()
    torch.manual_seed(99)
    torch.randint, torch.randn_like, torch.rand = randint, randn_like, rand
    try:
        model.eval()
        with torch.no_grad():
            vl = model(z0, None, ["a"] * B, validation_mode=2, run_teacher=True)
    finally:
        torch.randint, torch.randn_like, torch.rand = o_randint, o_randn_like, o_rand
    out["val_losses"] = np.array([float(v) for v in vl])
    out["val_noise"] = rec["randn_like"].numpy()
    out["val_guidance"] = rec["rand"].numpy() * 6

    # ---- _query_teacher alone
    with torch.no_grad():
        ts = model.noise_scheduler.timesteps[torch.tensor([0, 6, 32])]
        zq = cases.t(spec.det_uniform("distill.zq", (B, 8, H, W), 15))
        w = torch.tensor([0.5, 3.0, 5.5])
        out["query_teacher"] = model._query_teacher(zq, ts, P["embeds_cf"], P["mask_cf"], w).numpy()

    # ---- inference: 1-step, 2-step with post-CFG (noise fixed by seed), teacher Heun loop
    sched = ref_import.make_heun(ns)
    lat_shape = (B, 8, 256, 16)
    noise = cases.t(spec.det_uniform("distill.inf_noise", lat_shape, 16)) * np.float32(np.sqrt(3.0))
    import diffusers.utils as DU
    import models.audio_consistency_model as ACM
    ACM.randn_tensor = lambda shape, generator=None, device=None, dtype=None: noise.clone()
    with torch.no_grad():
        model.eval()
        stu1, tea, _, _ = model.inference(["a"] * B, sched, guidance_scale_input=4.0, guidance_scale_post=1.0,
                                          num_steps=1, use_edm=True, use_ema=True, query_teacher=True,
                                          num_teacher_steps=3, return_all=True)
        out["inf_student_1step"] = stu1.numpy()
        out["inf_teacher_3steps"] = tea.numpy()
        torch.manual_seed(7)
        rec.clear()
        torch.randn_like = randn_like
        try:
            stu2 = model.inference(["a"] * B, sched, guidance_scale_input=3.0, guidance_scale_post=2.0, num_steps=2,
                                   use_edm=True, use_ema=False)
        finally:
            torch.randn_like = o_randn_like
        out["inf_student_2step_cfg"] = stu2.numpy()
        out["inf_renoise"] = rec["randn_like"].numpy()

    # ---- EMA
    model.train()
    before_t = {k: v.clone() for k, v in model.student_target_unet.state_dict().items()}
    with torch.no_grad():
        model.update_ema()
    k0 = "down_blocks.1.resnets.0.conv1.weight"
    out["ema_target_after"] = model.student_target_unet.state_dict()[k0].numpy()[:4]
    out["ema_ema_after"] = model.student_ema_unet.state_dict()[k0].numpy()[:4]
    out["ema_key"] = np.array(k0)
    path = os.path.join(HERE, "distill_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    print({k: (v if np.ndim(v) == 0 or np.size(v) < 8 else np.shape(v)) for k, v in out.items()})


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    main()
