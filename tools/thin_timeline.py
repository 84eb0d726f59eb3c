#!/usr/bin/env python3
"""Per-workgroup phase timeline of one 2-D conv_gemm launch (ctta_conv_debug_stamps), for the thin K-heavy shapes of the
distillation step: prologue (entry -> first K tile landed), main loop, epilogue, per-K-step time, workgroups per CU.
usage: thin_timeline.py B H W Cin Cout k [variant] [cold=1]"""
import ctypes
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

B, H, W, Cin, Cout, k = [int(v) for v in sys.argv[1:7]]
var = int(sys.argv[7]) if len(sys.argv) > 7 else 0
cold = int(sys.argv[8]) if len(sys.argv) > 8 else 1
L = N.lib()
x = (torch.randn(B, H, W, Cin, device="cuda:0") * 0.5).to(torch.bfloat16)
out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda:0")
bias = torch.randn(Cout, device="cuda:0")
K = k * k * Cin
k_pad = (K + 63) // 64 * 64
ncopy = max(2, (600 << 20) // (Cout * k_pad * 2) + 1) if cold else 1
ws = [(torch.randn(Cout, k_pad, device="cuda:0") * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
d = N.ConvDesc()
d.x0, d.c0 = x.data_ptr(), Cin
d.batch, d.hi, d.wi, d.ho, d.wo = B, H, W, H, W
d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = k, k, 1, 1, 1, 1
d.pad_h, d.pad_w = (k - 1) // 2, (k - 1) // 2
d.w, d.k_pad, d.n, d.bias = ws[0].data_ptr(), k_pad, Cout, bias.data_ptr()
d.alpha, d.groups, d.out, d.ldc, d.tile = 1.0, 1, out.data_ptr(), Cout, var
for i in range(ncopy):
    d.w = ws[i].data_ptr()
    N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
torch.cuda.synchronize()
# timing without stamps, weights rotating
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(2 * ncopy):
    d.w = ws[i % ncopy].data_ptr()
    N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
e1.record()
torch.cuda.synchronize()
M = B * H * W
ms = e0.elapsed_time(e1) / (2 * ncopy)
print("M %d N %d K %d variant %d cold %d: %.1f us per launch = %.0f TFLOP/s" % (M, Cout, K, var, cold, ms * 1e3, 2.0 * M * Cout * K / ms / 1e9))
nwg = 1 << 16
buf = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda:0")
L.ctta_conv_debug_stamps(buf.data_ptr())
d.w = ws[0].data_ptr()
e0.record()
N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
e1.record()
torch.cuda.synchronize()
L.ctta_conv_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 6)
s = s[s[:, 1] != 0]
hw = s[:, 0] & 0xFFFFFFFF
xcc = (s[:, 0] >> 32) & 0xF
cu_key = (xcc << 16) | (hw & 0xFF00)
t = s[:, 1:5].astype(np.float64)
t0 = t[:, 0].min()
MHz = float(os.environ.get("TICK_MHZ", "100"))     # s_memtime: 100 MHz constant clock on this part
us = lambda v: v / MHz
print("stamped launch %.1f us by events; %d workgroups on %d CUs; span by stamps %.1f us"
      % (e0.elapsed_time(e1) * 1e3, len(s), len(set(cu_key.tolist())), us(t[:, 3].max() - t0)))
pro, main, epi = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
for name, v in (("entry (after first)", us(t[:, 0] - t0)), ("prologue", pro), ("main loop", main), ("epilogue", epi)):
    print("  %-20s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f" % (name, v.mean(), *np.percentile(v, [10, 50, 90]), v.max()))
by = defaultdict(list)
for i, kk in enumerate(cu_key.tolist()):
    by[kk].append(i)
per_cu = [len(v) for v in by.values()]
per_xcc = defaultdict(int)
for v in xcc.tolist():
    per_xcc[v] += 1
print("  workgroups per CU: min %d max %d mean %.2f; per XCC %s" % (min(per_cu), max(per_cu), np.mean(per_cu), dict(sorted(per_xcc.items()))))
