#!/usr/bin/env python3
"""GroupNorm(+SiLU) microbenchmark through the C ABI (three-pass form: partial sums, finalize, apply) at the B = 32 shapes of
the VAE decoder and the U-Net; ms per call and GB/s against the 3 tensor passes it moves (2 reads + 1 write)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def main():
    L = N.lib()
    B = 32
    for hw, c in ((65536, 128), (65536, 256), (16384, 256), (16384, 512), (4096, 512), (4096, 320), (1024, 640), (256, 1280)):
        x = torch.randn(B, hw, c, device=DEV).to(torch.bfloat16)
        y = torch.empty_like(x)
        g, b = torch.randn(c, device=DEV), torch.randn(c, device=DEV)
        scratch = torch.empty(L.ctta_groupnorm_scratch_floats(B, hw, c, 32), device=DEV)
        s = N.stream_ptr()

        def run():
            N.check(L.ctta_groupnorm(N.ptr(x), N.ptr(y), B, hw, c, 32, N.ptr(g), N.ptr(b), 1e-6, 1, N.ptr(scratch), s))
        run()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ts = []
        for _ in range(5):
            e[0].record()
            for _ in range(5):
                run()
            e[1].record()
            torch.cuda.synchronize()
            ts.append(e[0].elapsed_time(e[1]) / 5)
        ms = sorted(ts)[2]
        print("hw %6d c %4d: %.3f ms  %.0f GB/s over 3 passes" % (hw, c, ms, 3 * x.numel() * 2 / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
