#!/usr/bin/env python3
"""LayerNorm backward microbenchmark through the C ABI at the distillation step's token-matrix shapes (batch 9):
us per call and GB/s over the three tensor passes (x, dy read; dx written).  (Until round 5 an environment knob selected the
parameter-gradient reduction (global atomics / partial table + fixed-order fold)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def main():
    L = N.lib()
    for rows, d, ld in ((36864, 255, 256), (36864, 320, 320), (9216, 510, 512), (9216, 640, 640), (2304, 1020, 1024), (2304, 1280, 1280)):
        x = torch.randn(rows, ld, device=DEV).to(torch.bfloat16)
        dy = torch.randn(rows, ld, device=DEV).to(torch.bfloat16)
        dx = torch.empty_like(x)
        g = torch.randn(d, device=DEV)
        dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        s = N.stream_ptr()

        def run():
            N.check(L.ctta_layernorm_bwd(N.ptr(x), N.ptr(dy), N.ptr(dx), rows, d, ld, N.ptr(g), 1e-5, 0, N.ptr(dg), N.ptr(db), s))
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
        ms = sorted(ts)[2]
        print("rows %6d d %4d: %.1f us  %.0f GB/s over 3 passes" % (rows, d, ms * 1e3, 3 * x.numel() * 2 / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
