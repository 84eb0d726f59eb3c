#!/usr/bin/env python3
"""Where do the device-to-device copies of an eager distillation step come from?  torch.profiler over two `train_step`
calls (batch 9, the bench's setup), Memcpy DtoD / copy kernels grouped by the Python frame that issued them."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import spec  # noqa: E402
from consistencytta_amd.models import AudioLCM  # noqa: E402
from consistencytta_amd.optim import WarmupSchedule  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, L = 9, 32
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tango_diffusion_light.json", unet_config=spec.LIGHT_UNET_CONFIG, snr_gamma=5.0,
                 use_edm=True, teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse",
                 target_ema_decay=0.95, ema_decay=0.999)
    m.to(dev)
    m.teacher_unet.init_random_(seed=10)
    m.student_unet.init_random_(seed=11)
    m.train()
    opt = m.prepare_training(lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, broadcast=True)
    sched = WarmupSchedule(opt, "linear", num_warmup_steps=1000, num_training_steps=100000)
    g = torch.Generator(device="cpu").manual_seed(5)
    z0 = (torch.randn(B, 8, 256, 16, generator=g) * 0.9).to(dev)
    enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
    mask = torch.ones(B, L, dtype=torch.bool, device=dev)
    unc, umask = torch.zeros_like(enc), torch.zeros_like(mask)
    umask[:, 0] = True
    P = {"embeds_cf": torch.cat([unc, enc]), "mask_cf": torch.cat([umask, mask]), "embeds": enc, "mask": mask}
    for _ in range(2):
        m.train_step(z0, P, opt, sched)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            m.train_step(z0, P, opt, sched)
        torch.cuda.synchronize()
    by = collections.Counter()
    tot = collections.Counter()
    for e in prof.events():
        name = e.name
        if not (name.startswith("aten::copy_") or name.startswith("aten::clone") or name.startswith("aten::contiguous") or
                name.startswith("aten::to") or name.startswith("aten::where") or name.startswith("aten::cat")):
            continue
        frame = next((s for s in (e.stack or []) if "consistencytta_amd" in s or "bench.py" in s), "?")
        by[(name, frame)] += 1
        tot[name] += 1
    print("per 2 steps:", dict(tot))
    for (name, frame), n in by.most_common(25):
        print("%5d  %-18s %s" % (n, name, frame))


if __name__ == "__main__":
    main()
