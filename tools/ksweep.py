#!/usr/bin/env python3
"""Launch time of one conv_gemm shape as a function of K (kernel width), per tile variant: the intercept of the fit is
the per-launch cost that does not scale with K (prologue + epilogue of every tile), the slope the main-loop rate.
usage: ksweep.py C L [variants...]   (1-D conv, batch 32, C -> C channels, length L)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

C, Lw = int(sys.argv[1]), int(sys.argv[2])
variants = [int(v) for v in sys.argv[3:]] or [0]
L = N.lib()
B = 32
x = (torch.randn(B, 1, Lw, C, device="cuda:0") * 0.5).to(torch.bfloat16)
out = torch.empty(B, 1, Lw, C, dtype=torch.bfloat16, device="cuda:0")
bias = torch.randn(C, device="cuda:0")
for var in variants:
    pts = []
    for kw in (1, 2, 3, 5, 7, 11, 15):
        K = kw * C
        k_pad = (K + 63) // 64 * 64
        w = (torch.randn(C, k_pad, device="cuda:0") * 0.05).to(torch.bfloat16)
        d = N.ConvDesc()
        d.x0, d.c0 = x.data_ptr(), C
        d.batch, d.hi, d.wi, d.ho, d.wo = B, 1, Lw, 1, Lw
        d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = 1, kw, 1, 1, 1, 1
        d.pad_h, d.pad_w = 0, (kw - 1) // 2
        d.w, d.k_pad, d.n, d.bias = w.data_ptr(), k_pad, C, bias.data_ptr()
        d.alpha, d.groups, d.out, d.ldc, d.tile = 1.0, 1, out.data_ptr(), C, var
        if os.environ.get("KS_NOSTORE") == "1":   # experiment: the epilogue runs but (almost) nothing is stored
            d.out_limit = 4
        if kw % 2 == 0:
            d.wo = Lw   # even widths: still "same" length with the left-biased padding
        if L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()) != 0:
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        pts.append((K, e0.elapsed_time(e1) / 10 * 1e3))
    if len(pts) >= 2:
        n = len(pts)
        sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
        sxx = sum(p[0] * p[0] for p in pts); sxy = sum(p[0] * p[1] for p in pts)
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
        icpt = (sy - slope * sx) / n
        M = B * Lw
        print("variant %2d  M=%d N=%d: " % (var, M, C) + " ".join("K=%d:%.0fus" % p for p in pts))
        print("            intercept %.0f us, slope %.3f us per K  -> main-loop rate %.0f TFLOP/s, output+input bytes %.0f MB"
              % (icpt, slope, 2.0 * M * C / slope / 1e6, M * C * 4 / 1e6))
