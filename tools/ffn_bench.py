#!/usr/bin/env python3
"""Fused GEGLU feed-forward (ctta_ffn_geglu, csrc/ffn_fused.hip) against the two conv_gemm launches it replaces
(ff1 with the out_act = 4 epilogue, ff2 with the residual epilogue) through the C ABI: bit comparison and TFLOP/s
(2 * M * cp * 3 * ffp flops) at the level-0 token counts of generation (M = 131 072) and the teacher (36 864)."""
import ctypes
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from consistencytta_amd import _native as N  # noqa: E402
from gpu_util import conv_desc, pack_conv_weight  # noqa: E402

DEV = "cuda:0"


def time_ms(fn, reps=10, rounds=5):
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(rounds):
        e[0].record()
        for _ in range(reps):
            fn()
        e[1].record()
        torch.cuda.synchronize()
        ts.append(e[0].elapsed_time(e[1]) / reps)
    return sorted(ts)[len(ts) // 2]


def main():
    L = N.lib()
    cp, ffp = 256, 1024
    g = torch.Generator().manual_seed(1)
    w1 = (torch.randn(2 * ffp, cp, generator=g) * (1.5 / math.sqrt(cp)))
    b1 = torch.randn(2 * ffp, generator=g) * 0.2
    w2 = torch.randn(cp, ffp, generator=g) * (1.0 / math.sqrt(ffp))
    b2 = torch.randn(cp, generator=g) * 0.1
    w1p, k1 = pack_conv_weight(w1[:, :, None, None])
    w2p, k2 = pack_conv_weight(w2[:, :, None, None])
    packed = torch.empty(L.ctta_ffn_pack_bytes(cp, ffp), dtype=torch.uint8, device=DEV)
    N.check(L.ctta_ffn_pack(N.ptr(w1p), k1, N.ptr(w2p), k2, cp, ffp, N.ptr(packed), N.stream_ptr()))
    b1d, b2d = b1.to(DEV), b2.to(DEV)
    st = N.stream_ptr()
    for M in ([int(a) for a in sys.argv[1:]] or [131072, 36864, 300]):
        x = (torch.randn(M, cp, generator=g)).to(torch.bfloat16).to(DEV)
        res = (torch.randn(M, cp, generator=g)).to(torch.bfloat16).to(DEV)
        gg = torch.empty(M, ffp, dtype=torch.bfloat16, device=DEV)
        out_a = torch.empty(M, cp, dtype=torch.bfloat16, device=DEV)
        out_b = torch.full((M, cp), 7.0, dtype=torch.bfloat16, device=DEV)
        d1 = conv_desc(x0=x, c0=cp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w1p, k_pad=k1, n=2 * ffp, bias=b1d, out=gg, ldc=ffp, out_act=4)
        d2 = conv_desc(x0=gg, c0=ffp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w2p, k_pad=k2, n=cp, bias=b2d, res=res, res_ld=cp,
                       out=out_a, ldc=cp)

        def two():
            N.check(L.ctta_conv_gemm(ctypes.byref(d1), st))
            N.check(L.ctta_conv_gemm(ctypes.byref(d2), st))

        def one():
            N.check(L.ctta_ffn_geglu(N.ptr(x), cp, M, cp, ffp, N.ptr(packed), N.ptr(b1d), N.ptr(b2d), N.ptr(res), cp, N.ptr(out_b),
                                     cp, cp, st))
        two()
        one()
        torch.cuda.synchronize()
        same = torch.equal(out_a, out_b)
        diff = float((out_a.float() - out_b.float()).abs().max())
        fl = 2.0 * M * cp * 3 * ffp
        t2, t1 = time_ms(two), time_ms(one)
        print("M %7d: two launches %.3f ms (%.0f TF/s)   fused %.3f ms (%.0f TF/s)   bit-identical %s (max |diff| %.3g)"
              % (M, t2, fl / t2 / 1e9, t1, fl / t1 / 1e9, same, diff), flush=True)


if __name__ == "__main__":
    main()
