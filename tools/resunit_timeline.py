#!/usr/bin/env python3
"""Per-workgroup phase timeline of one fused ResBlock-unit launch (ctta_conv_debug_stamps): staging, conv1, intermediate
write, conv2, epilogue, and the gap to the next workgroup's entry on the same CU.
usage: resunit_timeline.py C K DIL [batch]      at the vocoder's stage length for C (128: 40960, 64: 81920, 32: 163840)"""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

C, k, dil = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
Lw = {128: 40960, 64: 81920, 32: 163840}[C]
L = N.lib()
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, Lw, C, generator=g) * 0.5).to(torch.bfloat16).to("cuda:0")
out = torch.empty_like(x)
w1, w2 = [(torch.randn(C * k * C, generator=g) * (C * k) ** -0.5).to(torch.bfloat16).to("cuda:0") for _ in range(2)]
b1, b2 = [(torch.randn(C, generator=g) * 0.1).to("cuda:0") for _ in range(2)]


def run():
    N.check(L.ctta_resunit_conv1d(N.ptr(x), B, Lw, C, k, dil, N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2), 0.1, N.ptr(out), 0,
                                  1.0, 0.0, N.stream_ptr()))


for _ in range(3):
    run()
torch.cuda.synchronize()
buf = torch.zeros((1 << 16) * 8, dtype=torch.int64, device="cuda:0")
L.ctta_conv_debug_stamps(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
L.ctta_conv_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 1] != 0]
hw = s[:, 0] & 0xFFFFFFFF
xcc = (s[:, 0] >> 32) & 0xF
cu_key = (xcc << 16) | (hw & 0xFF00)
t = s[:, 1:7].astype(np.float64)
MHz = float(os.environ.get("TICK_MHZ", "100"))
us = lambda v: v / MHz
t0 = t[:, 0].min()
print("C%d k%d d%d B%d: launch %.1f us by events; %d workgroups on %d (xcc, cu) keys; span by stamps %.1f us"
      % (C, k, dil, B, e0.elapsed_time(e1) * 1e3, len(s), len(set(cu_key.tolist())), us(t[:, 5].max() - t0)))
names = ("stage", "conv1", "mid write", "conv2", "epilogue")
for i, name in enumerate(names):
    v = us(t[:, i + 1] - t[:, i])
    print("  %-10s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (name, v.mean(), *np.percentile(v, [10, 50, 90])))
v = us(t[:, 5] - t[:, 0])
print("  %-10s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % ("whole", v.mean(), *np.percentile(v, [10, 50, 90])))
by = defaultdict(list)
for i, key in enumerate(cu_key.tolist()):
    by[key].append(i)
conc = []
for key, idx in by.items():
    ev = sorted([(t[i, 0], 1) for i in idx] + [(t[i, 5], -1) for i in idx])
    cur, last, area = 0, ev[0][0], 0.0
    for tt, d in ev:
        area += cur * (tt - last)
        last, cur = tt, cur + d
    conc.append(area / max(ev[-1][0] - ev[0][0], 1))
print("  workgroups per CU: min %d max %d; mean resident workgroups per CU (entry .. last store issued) %.2f"
      % (min(len(v) for v in by.values()), max(len(v) for v in by.values()), float(np.mean(conc))))
