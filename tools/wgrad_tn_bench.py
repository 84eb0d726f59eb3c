#!/usr/bin/env python3
"""ctta_wgrad_tn microbenchmark (isolated): us per launch against rows M, output N x C and position splits S -- how much of a
launch is per-chunk work and how much is fixed (prologue, slab tile store)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def main():
    L = N.lib()
    st = N.stream_ptr()
    for (M, Nn, C) in ((36864, 256, 256), (36864, 256, 1024), (9216, 512, 512), (9216, 512, 2048), (2304, 1024, 1024), (2304, 1024, 4096), (36864, 2048, 256)):
        x = torch.randn(M, C, device=DEV).to(torch.bfloat16)
        dy = torch.randn(M, Nn, device=DEV).to(torch.bfloat16)
        ld = (C + 1 + 3) // 4 * 4
        for S in (1, 4, 8, 16, 32, 64):
            if M // (2 * S) < 64:
                continue
            mp = (M + 64 * S - 1) // (64 * S) * (64 * S)
            slabs = torch.empty(S, Nn, ld, device=DEV)

            def run():
                N.check(L.ctta_wgrad_tn(N.ptr(dy), Nn, Nn, N.ptr(x), C, C, M, mp, S, C, N.ptr(slabs), Nn * ld, ld, st))
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
            tiles = ((Nn + 127) // 128) * ((C + 127) // 128)
            print("M %6d N %5d C %5d S %2d: %7.1f us  %4d workgroups x %3d chunks  %6.1f TFLOP/s" % (
                M, Nn, C, S, us, tiles * S, mp // S // 64, 2.0 * M * Nn * C / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
