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


def time_many(fns, reps=20, rounds=7, warm=40):
    """median ms per call of each fn, the candidates alternating inside every round (the first-timed candidate of a cold
    device otherwise reads 5-15 % slow)."""
    for _ in range(warm):
        fns[0]()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for k, fn in enumerate(fns):
            fn()
            e[0].record()
            for _ in range(reps):
                fn()
            e[1].record()
            torch.cuda.synchronize()
            ts[k].append(e[0].elapsed_time(e[1]) / reps)
    return [sorted(t)[len(t) // 2] for t in ts]


def main():
    L = N.lib()
    cp = int(os.environ.get("FFN_CP", "256"))
    ffp, d = 4 * cp, cp - cp // 256       # 255 -> 256, 510 -> 512: the U-Net's inner widths (heads x 51) and their padding
    g = torch.Generator().manual_seed(1)
    w1 = (torch.randn(2 * ffp, cp, generator=g) * (1.5 / math.sqrt(cp)))
    w1[:, d:] = 0
    b1 = torch.randn(2 * ffp, generator=g) * 0.2
    w2 = torch.randn(cp, ffp, generator=g) * (1.0 / math.sqrt(ffp))
    b2 = torch.randn(cp, generator=g) * 0.1
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(d, generator=g)).to(DEV)
    w1p, k1 = pack_conv_weight(w1[:, :, None, None])
    w2p, k2 = pack_conv_weight(w2[:, :, None, None])
    packed = torch.empty(L.ctta_ffn_pack_bytes(cp, ffp), dtype=torch.uint8, device=DEV)
    N.check(L.ctta_ffn_pack(N.ptr(w1p), k1, N.ptr(w2p), k2, cp, ffp, N.ptr(packed), N.stream_ptr()))
    b1d, b2d = b1.to(DEV), b2.to(DEV)
    st = N.stream_ptr()
    fl_row = 2.0 * cp * 3 * ffp
    for M in ([int(a) for a in sys.argv[1:]] or [131072, 73728, 36864, 300]):
        s2 = (torch.randn(M, cp, generator=g)).to(torch.bfloat16).to(DEV)
        s2[:, d:] = 0
        n3 = torch.empty_like(s2)
        gg = torch.empty(M, ffp, dtype=torch.bfloat16, device=DEV)
        out_a = torch.empty(M, cp, dtype=torch.bfloat16, device=DEV)
        out_b = torch.full((M, cp), 7.0, dtype=torch.bfloat16, device=DEV)
        out_c = torch.full((M, cp), 7.0, dtype=torch.bfloat16, device=DEV)
        d1 = conv_desc(x0=n3, c0=cp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w1p, k_pad=k1, n=2 * ffp, bias=b1d, out=gg, ldc=ffp, out_act=4)
        d2 = conv_desc(x0=gg, c0=ffp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w2p, k_pad=k2, n=cp, bias=b2d, res=s2, res_ld=cp,
                       out=out_a, ldc=cp)

        def ln():
            N.check(L.ctta_layernorm(N.ptr(s2), N.ptr(n3), M, d, cp, N.ptr(gamma), N.ptr(beta), 1e-5, st))

        def two():
            N.check(L.ctta_conv_gemm(ctypes.byref(d1), st))
            N.check(L.ctta_conv_gemm(ctypes.byref(d2), st))

        def three():
            ln()
            two()

        def one(out=out_b):
            N.check(L.ctta_ffn_geglu(N.ptr(n3), cp, M, cp, ffp, N.ptr(packed), N.ptr(b1d), N.ptr(b2d), N.ptr(s2), cp, N.ptr(out),
                                     cp, cp, None, None, 0, 0.0, st))

        def one_ln(out=out_c):
            N.check(L.ctta_ffn_geglu(N.ptr(s2), cp, M, cp, ffp, N.ptr(packed), N.ptr(b1d), N.ptr(b2d), N.ptr(s2), cp, N.ptr(out),
                                     cp, cp, N.ptr(gamma), N.ptr(beta), d, 1e-5, st))

        def rows(bm, f):
            def r():
                L.ctta_ffn_debug_rows(bm)
                f()
                L.ctta_ffn_debug_rows(0)
            return r
        # the whole tail of a transformer block: attn2.to_out + residual, norm3, ff1, ff2 + residual, proj_out + residual
        hp = cp // 4 * 5
        w0 = torch.randn(cp, hp, generator=g) * (1.0 / math.sqrt(hp))
        w0p, k0 = pack_conv_weight(w0[:, :, None, None], k_mult=32)
        wp = torch.randn(cp, cp, generator=g) * (1.0 / math.sqrt(cp))
        w3p, k3 = pack_conv_weight(wp[:, :, None, None])
        fstream = torch.empty(L.ctta_ffn_proj_pack_bytes(cp, hp), dtype=torch.uint8, device=DEV)
        pstream = torch.empty(L.ctta_ffn_proj_pack_bytes(cp, cp), dtype=torch.uint8, device=DEV)
        N.check(L.ctta_ffn_proj_pack(N.ptr(w0p), k0, hp, cp, N.ptr(fstream), st))
        N.check(L.ctta_ffn_proj_pack(N.ptr(w3p), k3, cp, cp, N.ptr(pstream), st))
        att = torch.randn(M, hp, generator=g).to(torch.bfloat16).to(DEV)
        s1 = torch.randn(M, cp, generator=g).to(torch.bfloat16).to(DEV)
        xin = torch.randn(M, cp, generator=g).to(torch.bfloat16).to(DEV)
        s2b, s3b = torch.empty_like(s1), torch.empty_like(s1)
        o5, o3, o1 = torch.empty_like(s1), torch.empty_like(s1), torch.empty_like(s1)
        d0 = conv_desc(x0=att, c0=hp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w0p, k_pad=k0, n=cp, bias=b2d, res=s1, res_ld=cp, out=s2b, ldc=cp)
        d1b = conv_desc(x0=n3, c0=cp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w1p, k_pad=k1, n=2 * ffp, bias=b1d, out=gg, ldc=ffp, out_act=4)
        d2b = conv_desc(x0=gg, c0=ffp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w2p, k_pad=k2, n=cp, bias=b2d, res=s2b, res_ld=cp, out=s3b, ldc=cp)
        d3 = conv_desc(x0=s3b, c0=cp, batch=1, hi=M, wi=1, ho=M, wo=1, w=w3p, k_pad=k3, n=cp, bias=b2d, res=xin, res_ld=cp, out=o5, ldc=cp)

        def fdesc(front, tail, out):
            fd = N.FfnDesc()
            L.ctta_ffn_desc_init(ctypes.byref(fd))
            fd.M, fd.cp, fd.ffp = M, cp, ffp
            fd.packed, fd.b1, fd.b2 = packed.data_ptr(), b1d.data_ptr(), b2d.data_ptr()
            fd.x, fd.ld_x, fd.res, fd.res_ld = n3.data_ptr(), cp, s2b.data_ptr(), cp
            if front:
                fd.front_packed, fd.front_bias, fd.att, fd.att_ld, fd.front_k = fstream.data_ptr(), b2d.data_ptr(), att.data_ptr(), hp, hp
                fd.front_res, fd.front_res_ld, fd.s2_out, fd.s2_ld = s1.data_ptr(), cp, s2b.data_ptr(), cp
                fd.ln_gamma, fd.ln_beta, fd.ln_d, fd.ln_eps = gamma.data_ptr(), beta.data_ptr(), d, 1e-5
            if tail:
                fd.proj_packed, fd.proj_bias, fd.proj_res, fd.proj_res_ld = pstream.data_ptr(), b2d.data_ptr(), xin.data_ptr(), cp
            fd.out, fd.ldc, fd.n_valid = out.data_ptr(), cp, cp
            return fd
        fd_t, fd_ft = fdesc(False, True, o3), fdesc(True, True, o1)

        def ln_b():
            N.check(L.ctta_layernorm(N.ptr(s2b), N.ptr(n3), M, d, cp, N.ptr(gamma), N.ptr(beta), 1e-5, st))

        def five():
            N.check(L.ctta_conv_gemm(ctypes.byref(d0), st))
            ln_b()
            N.check(L.ctta_conv_gemm(ctypes.byref(d1b), st))
            N.check(L.ctta_conv_gemm(ctypes.byref(d2b), st))
            N.check(L.ctta_conv_gemm(ctypes.byref(d3), st))

        def three_t():
            N.check(L.ctta_conv_gemm(ctypes.byref(d0), st))
            ln_b()
            N.check(L.ctta_ffn_block(ctypes.byref(fd_t), st))

        def one_ft():
            N.check(L.ctta_ffn_block(ctypes.byref(fd_ft), st))
        five()
        three_t()
        one_ft()
        torch.cuda.synchronize()
        fl5 = fl_row + 2.0 * cp * (hp + cp)
        tb = time_many([five, three_t, one_ft])
        print("cp %d  M %d block tail (to_out, norm3, ff1, ff2, proj_out): all equal %s" % (cp, M, torch.equal(o5, o3) and torch.equal(o5, o1)))
        for name, t in zip(("five launches", "to_out + norm3 + fused(ffn, proj_out)", "one launch (front + ffn + tail)"), tb):
            print("   %-40s %.4f ms  %.0f TF/s" % (name, t, fl5 * M / t / 1e9), flush=True)
        three()
        one()
        one_ln()
        torch.cuda.synchronize()
        same = torch.equal(out_a, out_b)
        same_ln = torch.equal(out_a, out_c)
        diff = float((out_a.float() - out_c.float()).abs().max())
        cands = [("LN + ff1 + ff2 (3 launches)", three), ("ff1 + ff2 (2 launches)", two), ("fused", one), ("fused + LN on load", one_ln)]
        for bm in ((128, 144) if cp == 256 else (48, 64, 80)):
            cands.append(("fused, %d-row tile" % bm, rows(bm, one)))
        tt = time_many([c[1] for c in cands])
        print("cp %d  M %d (fused wanted: %d): fused == two launches: %s; fused with LayerNorm on load == three launches: %s (max |diff| %.3g)"
              % (cp, M, L.ctta_ffn_geglu_wanted(cp, ffp, M), same, same_ln, diff))
        for (name, _), t in zip(cands, tt):
            print("   %-30s %.4f ms  %.0f TF/s" % (name, t, fl_row * M / t / 1e9), flush=True)


if __name__ == "__main__":
    main()
