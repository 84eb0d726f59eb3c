#!/usr/bin/env python3
"""Runs ONE conv_gemm shape/variant a few times (target for rocprofv3 --pmc passes).
usage: pmc_probe.py <shape index in sweep_conv.SHAPES> <variant id> [reps]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402
from sweep_conv import SHAPES  # noqa: E402

idx, var = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tag, B, H, W, Cin, Cout, kh, kw, dil = SHAPES[idx]
L = N.lib()
x = (torch.randn(B, H, W, Cin, device="cuda:0") * 0.5).to(torch.bfloat16)
K = kh * kw * Cin
k_pad = (K + 63) // 64 * 64
w = (torch.randn(Cout, k_pad, device="cuda:0") * 0.05).to(torch.bfloat16)
bias = torch.randn(Cout, device="cuda:0")
out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda:0")
d = N.ConvDesc()
d.x0, d.c0 = x.data_ptr(), Cin
d.batch, d.hi, d.wi, d.ho, d.wo = B, H, W, H, W
d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = kh, kw, 1, 1, 1, dil
d.pad_h, d.pad_w = (kh - 1) // 2, (kw - 1) * dil // 2
d.w, d.k_pad, d.n, d.bias = w.data_ptr(), k_pad, Cout, bias.data_ptr()
d.alpha, d.groups, d.out, d.ldc, d.tile = 1.0, 1, out.data_ptr(), Cout, var
for _ in range(reps):
    N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
torch.cuda.synchronize()
print(tag, "variant", var, "flops", 2.0 * B * H * W * Cout * K)
