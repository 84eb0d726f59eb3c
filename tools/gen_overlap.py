#!/usr/bin/env python3
"""Experiment (round 4): does running the three generation stages of DIFFERENT batches concurrently pay?  U-Net(i + 2),
VAE decoder(i + 1) and HiFi-GAN(i) as three hipGraphs on three streams per iteration (inputs handed over by small device
copies at the iteration boundary) against the same three graphs back to back on one stream."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402
from consistencytta_amd import modules, spec  # noqa: E402
from consistencytta_amd.models import ConsistencyTTA  # noqa: E402

dev = torch.device("cuda:0")
B, L = 32, 32
vae = modules.AutoencoderKL(ddconfig=spec.VAE_DDCONFIG, embed_dim=8, scale_factor=0.9227914214134216)
pipe = ConsistencyTTA(unet_config=spec.LIGHT_UNET_CONFIG, vae=vae)
pipe.to(dev)
pipe.unet.init_random_(seed=0)
vae.init_random_(seed=1)
pipe.eval().requires_grad_(False)
g = torch.Generator(device="cpu").manual_seed(3)
enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
lens = torch.randint(6, L + 1, (B,), generator=g)
mask = (torch.arange(L)[None, :] < lens[:, None]).to(dev)
noise = torch.randn(B, 8, 256, 16, generator=g).to(dev)
scratch = torch.empty(4, dtype=torch.float32, device=dev)


def graphed(fn):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        out = fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        out = fn()
    return gr, out


with torch.no_grad():
    gU, lat = graphed(lambda: pipe.generate_latent(enc, mask, noise, cfg_scale_input=4.0, cfg_scale_post=1.0, num_steps=1))
    lat_in = lat.clone()
    gV, mel = graphed(lambda: vae.decode_first_stage(lat_in))
    mel_in = mel.clone()

    def voc():
        wav = vae.vocode(mel_in)
        pcm = torch.empty(wav.shape, dtype=torch.int16, device=dev)
        N.check(N.lib().ctta_wav_finalize(N.ptr(wav), wav.numel(), N.ptr(scratch), None, N.ptr(pcm), N.stream_ptr()))
        return pcm
    gH, pcm = graphed(voc)
torch.cuda.synchronize()


def seq(n):
    for _ in range(n):
        gU.replay(); lat_in.copy_(lat); gV.replay(); mel_in.copy_(mel); gH.replay()


sU, sV, sH = (torch.cuda.Stream(device=dev) for _ in range(3))


def piped(n):
    cur = torch.cuda.current_stream(dev)
    for _ in range(n):
        lat_in.copy_(lat); mel_in.copy_(mel)
        for s_, g_ in ((sH, gH), (sV, gV), (sU, gU)):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                g_.replay()
        for s_ in (sU, sV, sH):
            cur.wait_stream(s_)


for name, fn in (("sequential", seq), ("3-stage pipeline", piped), ("sequential", seq), ("3-stage pipeline", piped)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(20)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-18s %.2f ms per batch of %d = %.1f clips/s" % (name, dt * 1e3, B, B / dt))
