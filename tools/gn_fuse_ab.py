#!/usr/bin/env python3
"""A/B of GroupNorm statistics from the conv epilogue, launch by launch (HIP events, interleaved rounds):
   conv (no partials) + 3-pass GroupNorm   vs   conv (partials in the straight-line epilogue) + finalize/apply.
Shapes are the 3x3 convs of the U-Net levels and the VAE decoder at batch 32."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from consistencytta_amd import _native as N  # noqa: E402
from gpu_util import conv_desc, DEV  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, with_res
    (32, 256, 16, 256, 256, True), (32, 128, 8, 512, 512, True), (32, 64, 4, 1024, 1024, True),
    (32, 256, 16, 512, 512, True), (32, 512, 32, 256, 256, True), (32, 1024, 64, 128, 128, True),
    (32, 256, 16, 256, 256, False),
]


def timed(fn, n=6):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    fn()
    torch.cuda.synchronize()
    e[0].record()
    for _ in range(n):
        fn()
    e[1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n * 1e3


def main():
    L_ = N.lib()
    G = 32
    for (B, H, W, C, Co, with_res) in SHAPES:
        hw = H * W
        x = torch.randn(B, H, W, C, device=DEV).to(torch.bfloat16)
        w = (torch.randn(Co, 9 * C, device=DEV) / (9 * C) ** 0.5).to(torch.bfloat16)
        bias = torch.randn(Co, device=DEV) * 0.1
        res = torch.randn(B, H, W, Co, device=DEV).to(torch.bfloat16)
        out = torch.empty(B, H, W, Co, dtype=torch.bfloat16, device=DEV)
        y = torch.empty_like(out)
        gamma, beta = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
        part = torch.zeros(B * (hw // 16 + 1) * G * 2, dtype=torch.float32, device=DEV)
        scratch = torch.empty(L_.ctta_groupnorm_scratch_floats(B, hw, Co, G) + B * 2 * Co, dtype=torch.float32, device=DEV)
        kw = dict(x0=x, c0=C, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=w, k_pad=9 * C, n=Co, bias=bias,
                  out=out, ldc=Co)
        if with_res:
            kw.update(res=res, res_ld=Co)
        d0 = conv_desc(**kw)
        d1 = conv_desc(gn_part=part, gn_groups=G, gn_hw=hw, gn_part_floats=part.numel(), **kw)
        s = N.stream_ptr()

        def conv0():
            N.check(L_.ctta_conv_gemm(ctypes.byref(d0), s))

        def conv1():
            N.check(L_.ctta_conv_gemm(ctypes.byref(d1), s))
        conv1()
        chunks = L_.ctta_conv_last_gn_chunks()

        def gn3():
            N.check(L_.ctta_groupnorm(N.ptr(out), N.ptr(y), B, hw, Co, G, N.ptr(gamma), N.ptr(beta), 1e-5, 1, N.ptr(scratch), s))

        def gnp():
            N.check(L_.ctta_groupnorm_from_partials(N.ptr(out), N.ptr(y), B, hw, Co, G, N.ptr(gamma), N.ptr(beta), 1e-5, 1,
                                                    N.ptr(part), chunks, N.ptr(scratch), None, s))

        def seq0():
            conv0(); gn3()

        def seq1():
            conv1(); gnp()
        r = {}
        for rnd in range(3):
            for name, fn in (("conv0", conv0), ("conv1", conv1), ("gn3", gn3), ("gnp", gnp), ("seq0", seq0), ("seq1", seq1)):
                r.setdefault(name, []).append(timed(fn))
        med = {k: sorted(v)[1] for k, v in r.items()}
        mb = B * hw * Co * 2 / 1e6
        print("B%d %dx%d %d->%d res=%d chunks=%d | conv %.1f -> %.1f us | gn 3-pass %.1f -> partials %.1f us (tensor %.0f MB) | "
              "conv+gn %.1f -> %.1f us" % (B, H, W, C, Co, with_res, chunks, med["conv0"], med["conv1"], med["gn3"], med["gnp"], mb,
                                           med["seq0"], med["seq1"]), flush=True)


if __name__ == "__main__":
    main()
