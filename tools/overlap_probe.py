#!/usr/bin/env python3
"""Does an HBM-streaming kernel (the two-shadow EMA pass over 559 M parameters, 2.4 ms alone) hide beside the pipelined
distillation step's GEMM-heavy graphs?  Times the pipelined step with 0 / 1 / 3 dummy EMA passes per step on a side stream
(launched when the step starts).  If the step grows by much less than the passes take alone, the optimizer tail (AdamW, EMA,
zero fill: ~5.6 ms alone at the end of every step) is worth moving beside the backward block by block."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402
from consistencytta_amd import spec  # noqa: E402
from consistencytta_amd.models import AudioLCM  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, L = 9, 32
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tango_diffusion_light.json", unet_config=spec.LIGHT_UNET_CONFIG, snr_gamma=5.0,
                 use_edm=True, teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse",
                 target_ema_decay=0.95, ema_decay=0.999)
    m.to(dev)
    m.teacher_unet.init_random_(seed=10)
    m.student_unet.init_random_(seed=11)
    with torch.no_grad():
        for dst in (m.student_target_unet, m.student_ema_unet):
            for p, q in zip(dst.parameters(), m.student_unet.parameters()):
                p.copy_(q)
    m.train()
    opt = m.prepare_training(lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, broadcast=False)
    g = torch.Generator(device="cpu").manual_seed(5)
    z0 = (torch.randn(B, 8, 256, 16, generator=g) * 0.9).to(dev)
    enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
    lens = torch.randint(6, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(dev)
    unc, umask = torch.zeros_like(enc), torch.zeros_like(mask)
    umask[:, 0] = True
    P = {"embeds_cf": torch.cat([unc, enc]), "mask_cf": torch.cat([umask, mask]), "embeds": enc, "mask": mask}
    kw = dict(time_inds=torch.randint(0, 17, (B,), generator=g) * 2, gaussian_noise=torch.randn(B, 8, 256, 16, generator=g).to(dev),
              guidance_scale=torch.rand(B, generator=g) * 6)
    gs = m.capture_train_graph(opt, z0, P, segmented=False, pipeline_teacher=True, **kw)
    n = opt.flat.numel()
    d0, d1, d2 = (torch.zeros(n, device=dev) for _ in range(3))
    side = torch.cuda.Stream(device=dev)
    L_ = N.lib()

    def dummy(k):
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(k):
                N.check(L_.ctta_ema_update2(N.ptr(d0), N.ptr(d1), 0.95, N.ptr(d2), 0.999, n, N.stream_ptr()))

    def run(k, steps=10):
        for _ in range(3):
            if k:
                dummy(k)
            gs.step(z0, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if k:
                dummy(k)
            gs.step(z0, None)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.cuda.stream(torch.cuda.current_stream(dev)):
        N.check(L_.ctta_ema_update2(N.ptr(d0), N.ptr(d1), 0.95, N.ptr(d2), 0.999, n, N.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    print("one EMA pass alone: %.2f ms" % e0.elapsed_time(e1))
    for k in (0, 1, 3, 0, 3):
        print("pipelined step with %d dummy EMA passes beside it: %.2f ms" % (k, run(k)), flush=True)


if __name__ == "__main__":
    main()
