#!/usr/bin/env python3
"""Per-workgroup phase timeline of one conv_gemm launch (ctta_conv_debug_stamps): where a tile's time goes -- launch gap
on its CU, prologue (entry -> first K tile landed), main loop, epilogue (until the last store is issued), and the gap
to the next workgroup's entry on the same CU (store drain + dispatch).
usage: tile_timeline.py C L KW [variant]      1-D conv, batch 32, C -> C channels, length L, kernel width KW"""
import ctypes
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

C, Lw, kw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
var = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L = N.lib()
B = 32
x = (torch.randn(B, 1, Lw, C, device="cuda:0") * 0.5).to(torch.bfloat16)
out = torch.empty(B, 1, Lw, C, dtype=torch.bfloat16, device="cuda:0")
bias = torch.randn(C, device="cuda:0")
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
for _ in range(3):
    N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
torch.cuda.synchronize()
nwg = 1 << 16
buf = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda:0")
L.ctta_conv_debug_stamps(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
e1.record()
torch.cuda.synchronize()
L.ctta_conv_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 6)
s = s[s[:, 1] != 0]
hw = s[:, 0] & 0xFFFFFFFF
xcc = (s[:, 0] >> 32) & 0xF
cu_key = (xcc << 16) | (hw & 0xFF00)          # XCC, SE/SH/CU fields of HW_ID (bits 8..15)
t = s[:, 1:5].astype(np.float64)
setup = (s[:, 5] - s[:, 1]).astype(np.float64)
t0 = t[:, 0].min()
MHz = float(os.environ.get('TICK_MHZ', '100'))   # s_memtime ticks: shader clock (about 1700 MHz under MFMA load here); default prints ticks / 100
us = lambda v: v / MHz
print("launch %.1f us by events; %d workgroups on %d distinct (xcc, cu) keys; span by stamps %.1f us"
      % (e0.elapsed_time(e1) * 1e3, len(s), len(set(cu_key.tolist())), us(t[:, 3].max() - t0)))
pro, main, epi = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2])
for name, v in (("setup", us(setup)), ("prologue", pro), ("main loop", main), ("epilogue", epi)):
    print("  %-10s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (name, v.mean(), *np.percentile(v, [10, 50, 90])))
by = defaultdict(list)
for i, k in enumerate(cu_key.tolist()):
    by[k].append(i)
gaps, per_cu = [], []
for k, idx in by.items():
    idx.sort(key=lambda i: t[i, 0])
    per_cu.append(len(idx))
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append(us(t[b, 0] - t[a, 3]))
gaps = np.array(gaps)
if len(gaps):
    print("  gap (epilogue issued -> next workgroup's entry on the same CU): mean %.2f us p10 %.2f p50 %.2f p90 %.2f"
          % (gaps.mean(), *np.percentile(gaps, [10, 50, 90])))
print("  workgroups per CU: min %d max %d; first entry spread %.2f us; last CU finishes %.1f us, earliest CU finishes %.1f us"
      % (min(per_cu), max(per_cu), us(np.percentile(t[:, 0], 1) - t0),
         us(max(t[i, 3] for i in range(len(s))) - t0), us(min(max(t[i, 3] for i in idx) for idx in by.values()) - t0)))
