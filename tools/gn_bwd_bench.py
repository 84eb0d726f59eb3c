#!/usr/bin/env python3
"""GroupNorm(+SiLU) backward microbenchmark through the C ABI at the batch-9 shapes of the distillation step (levels 1-3: the
single-launch small-slab kernel); us per call under rocprofv3-free HIP events, isolated (nothing else on the device)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def main():
    L = N.lib()
    B = 9
    for hw, c in ((1024, 512), (1024, 1024), (256, 1024), (256, 2048), (64, 1024), (64, 2048), (4096, 256)):
        x = torch.randn(B, hw, c, device=DEV).to(torch.bfloat16)
        dy = torch.randn(B, hw, c, device=DEV).to(torch.bfloat16)
        dx = torch.zeros_like(x)
        g, b = torch.randn(c, device=DEV), torch.randn(c, device=DEV)
        dg, db = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        stats = torch.empty(B, 32, 2, device=DEV)
        s = N.stream_ptr()
        N.check(L.ctta_groupnorm_stats(N.ptr(x), B, hw, c, 32, 1e-5, N.ptr(stats), s))
        scratch = torch.empty(L.ctta_groupnorm_bwd_scratch_floats(B, hw, c, 32), device=DEV)
        for acc in (0, 1):
            def run():
                N.check(L.ctta_groupnorm_bwd(N.ptr(x), N.ptr(dy), N.ptr(dx), B, hw, c, 32, N.ptr(stats), N.ptr(g), N.ptr(b), 1, acc,
                                             N.ptr(dg), N.ptr(db), 1, N.ptr(scratch), s))
            run()
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ts = []
            for _ in range(5):
                e[0].record()
                for _ in range(10):
                    run()
                e[1].record()
                torch.cuda.synchronize()
                ts.append(e[0].elapsed_time(e[1]) / 10)
            us = sorted(ts)[2] * 1e3
            nbytes = x.numel() * 2 * (4 if acc else 3)
            print("B 9 hw %5d c %4d acc %d: %6.1f us per call (both launches)  %.0f GB/s" % (hw, c, acc, us, nbytes / us / 1e3), flush=True)


if __name__ == "__main__":
    main()
