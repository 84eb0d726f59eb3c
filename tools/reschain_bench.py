#!/usr/bin/env python3
"""HiFi-GAN ResBlock microbenchmark through the C ABI: three ctta_resunit_conv1d launches vs one ctta_reschain_conv1d at the
vocoder's real stage shapes (B = 32; C = 64 at L = 81920, C = 32 at L = 163840); ms per ResBlock and algorithmic TFLOP/s."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def timeit(fn, inner=5, outer=5):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(outer):
        e[0].record()
        for _ in range(inner):
            fn()
        e[1].record()
        torch.cuda.synchronize()
        ts.append(e[0].elapsed_time(e[1]) / inner)
    return sorted(ts)[len(ts) // 2]


def main():
    L_ = N.lib()
    B = int(os.environ.get("RC_BATCH", "32"))
    dils = (1, 3, 5)
    dil_arr = (ctypes.c_int * 3)(*dils)
    for C, L in ((64, 81920), (32, 163840)):
        for k in (3, 7):
            g = torch.Generator().manual_seed(k)
            x = (torch.randn(B, L, C, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
            ws = [(torch.randn(C * k * C, generator=g) * (C * k) ** -0.5).to(torch.bfloat16).to(DEV) for _ in range(6)]
            bs = [(torch.randn(C, generator=g) * 0.1).to(DEV) for _ in range(6)]
            ta, tb, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            s = N.stream_ptr()

            def three():
                cur = x
                for u, d in enumerate(dils):
                    dst = out if u == 2 else (ta if u == 0 else tb)
                    N.check(L_.ctta_resunit_conv1d(N.ptr(cur), B, L, C, k, d, N.ptr(ws[2 * u]), N.ptr(bs[2 * u]),
                                                   N.ptr(ws[2 * u + 1]), N.ptr(bs[2 * u + 1]), 0.1, N.ptr(out if u == 2 else dst),
                                                   0, 1.0, 0.0, s))
                    cur = dst
            vp = lambda ts: (ctypes.c_void_p * 3)(*[N.ptr(t) for t in ts])
            args = (dil_arr, vp(ws[0::2]), vp(bs[0::2]), vp(ws[1::2]), vp(bs[1::2]))

            def chain():
                N.check(L_.ctta_reschain_conv1d(N.ptr(x), B, L, C, k, *args, 0.1, N.ptr(out), 0, 1.0, 0.0, s))
            fl = 3 * 2 * 2.0 * k * C * C * B * L
            line = "C%d k%d L%d: " % (C, k, L)
            t3 = timeit(three)
            line += "3 units %.3f ms (%.0f TF/s)" % (t3, fl / t3 / 1e9)
            if L_.ctta_reschain_supported(C, k, dil_arr):
                tc = timeit(chain)
                line += "  chain %.3f ms (%.0f TF/s)" % (tc, fl / tc / 1e9)
            print(line, flush=True)


if __name__ == "__main__":
    main()
