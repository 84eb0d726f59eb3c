#!/usr/bin/env python3
"""GroupNorm(+SiLU) apply pass (ctta_groupnorm_from_partials: statistics from the producing convolution, one read + one write of
the tensor) at the generation pipeline's shapes, through the C ABI: ms per call and GB/s over the two tensor passes.
(Round 6, one box: 5.4-6.1 TB/s = 0.68-0.76 of the HBM peak at every B = 32 shape; non-temporal loads / stores +6 % on the
537 MB / 1 GB tensors and -8 % on the 67 MB ones, 8 loads in flight and smaller grids no better: left as it is.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"
SHAPES = [("vae 1024x64 C256", 32, 65536, 256, 32), ("vae 1024x64 C128", 32, 65536, 128, 32), ("vae 512x32 C512", 32, 16384, 512, 32),
          ("vae 512x32 C256", 32, 16384, 256, 32), ("vae 256x16 C512", 32, 4096, 512, 32), ("unet 256x16 C256", 32, 4096, 256, 32),
          ("unet 256x16 C512 (concat)", 32, 4096, 512, 32), ("distill 256x16 C256 B9", 9, 4096, 256, 32)]


def main():
    L = N.lib()
    for tag, B, hw, c, G in SHAPES:
        x = torch.randn(B, hw, c, device=DEV).to(torch.bfloat16)
        y = torch.empty_like(x)
        nchunk = max(1, hw // 256)
        part = torch.zeros(B, nchunk, G, 2, device=DEV)
        part[..., 1] = 256.0 * c / G                      # sum of squares of a chunk: unit variance, zero mean
        gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        scratch = torch.empty(B * 2 * c + 1024, device=DEV)
        st = N.stream_ptr()

        def run():
            N.check(L.ctta_groupnorm_from_partials(N.ptr(x), N.ptr(y), B, hw, c, G, N.ptr(gamma), N.ptr(beta), 1e-6, 1, N.ptr(part), nchunk,
                                                   N.ptr(scratch), None, st))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ts = []
        for _ in range(7):
            e[0].record()
            for _ in range(10):
                run()
            e[1].record()
            torch.cuda.synchronize()
            ts.append(e[0].elapsed_time(e[1]) / 10)
        ms = sorted(ts)[3]
        line = "%-28s %.4f ms %5.0f GB/s" % (tag, ms, 2 * x.numel() * 2 / ms / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
