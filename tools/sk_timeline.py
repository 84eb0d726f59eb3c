#!/usr/bin/env python3
"""Per-workgroup phase timeline of one STREAM-K conv_gemm launch (ctta_conv_debug_stamps; conv_gemm_sk_kernel writes 8 words per
workgroup id): K steps and time of the contributor segment and of the owner segment(s), write-out of the partial tile, wait for
the partners, fold + epilogue.  Beside it the same shape on the one-tile-per-workgroup twin of the tile.
usage: sk_timeline.py B H W Cin Cout k [sk variant=41] [plain variant=29] [grid=0]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

B, H, W, Cin, Cout, k = [int(v) for v in sys.argv[1:7]]
var = int(sys.argv[7]) if len(sys.argv) > 7 else 41
plain = int(sys.argv[8]) if len(sys.argv) > 8 else 29
N.set_option("streamk_grid", int(sys.argv[9]) if len(sys.argv) > 9 else 0)
L = N.lib()
x = (torch.randn(B, H, W, Cin, device="cuda:0") * 0.5).to(torch.bfloat16)
out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda:0")
bias = torch.randn(Cout, device="cuda:0")
K = k * k * Cin
k_pad = (K + 63) // 64 * 64
ncopy = max(2, (600 << 20) // (Cout * k_pad * 2) + 1)
ws = [(torch.randn(Cout, k_pad, device="cuda:0") * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
d = N.ConvDesc()
d.x0, d.c0 = x.data_ptr(), Cin
d.batch, d.hi, d.wi, d.ho, d.wo = B, H, W, H, W
d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = k, k, 1, 1, 1, 1
d.pad_h, d.pad_w = (k - 1) // 2, (k - 1) // 2
d.w, d.k_pad, d.n, d.bias = ws[0].data_ptr(), k_pad, Cout, bias.data_ptr()
d.alpha, d.groups, d.out, d.ldc = 1.0, 1, out.data_ptr(), Cout
M = B * H * W
MHz = float(os.environ.get("TICK_MHZ", "100"))     # s_memtime: 100 MHz constant clock on this part


def timed(v):
    d.tile = v
    for i in range(ncopy):
        d.w = ws[i].data_ptr()
        N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(2 * ncopy):
        d.w = ws[i % ncopy].data_ptr()
        N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (2 * ncopy)
    print("M %d N %d K %d variant %d: %.1f us per launch = %.0f TFLOP/s" % (M, Cout, K, v, ms * 1e3, 2.0 * M * Cout * K / ms / 1e9))


def stats(name, v):
    v = np.asarray(v, dtype=np.float64)
    if len(v):
        print("  %-34s n %4d  mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f" % (name, len(v), v.mean(), *np.percentile(v, [10, 50, 90]), v.max()))


timed(plain)
buf = torch.zeros((1 << 16) * 6, dtype=torch.int64, device="cuda:0")
L.ctta_conv_debug_stamps(buf.data_ptr())
d.w = ws[0].data_ptr()
N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
torch.cuda.synchronize()
L.ctta_conv_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 6)
s = s[s[:, 1] != 0]
t = s[:, 1:5].astype(np.float64) / MHz
nk = (K + 63) // 64
print(" plain tile: %d workgroups, span %.1f us" % (len(s), t[:, 3].max() - t[:, 0].min()))
stats("prologue (us)", t[:, 1] - t[:, 0])
stats("main loop (us)", t[:, 2] - t[:, 1])
stats("   per K step (us)", (t[:, 2] - t[:, 1]) / nk)
stats("epilogue (us)", t[:, 3] - t[:, 2])

timed(var)
buf.zero_()
L.ctta_conv_debug_stamps(buf.data_ptr())
d.w = ws[0].data_ptr()
N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
torch.cuda.synchronize()
L.ctta_conv_debug_stamps(None)
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 1] != 0]
w0 = s[:, 0].astype(np.uint64)
sa = ((w0 >> np.uint64(36)) & np.uint64(0xfff)).astype(np.int64)
sb = ((w0 >> np.uint64(48)) & np.uint64(0xfff)).astype(np.int64)
tb = ((w0 >> np.uint64(60)) & np.uint64(0xf)).astype(np.int64)
xcc = ((w0 >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
t = s[:, 1:8].astype(np.float64) / MHz     # begin, later-part main end, its write-out end, first-part main end, its write-out end, K walk over, end
# (s_memtime is per XCD: only differences inside one workgroup mean anything)
print(" stream-K: %d workgroups (per XCC %s); K steps per workgroup: later parts %.1f + first parts / whole tiles %.1f" % (
    len(s), np.bincount(xcc, minlength=8).tolist(), sa.mean(), sb.mean()))
ca = sa > 0
stats("later part of a tile: main (ticks/100)", (t[ca, 1] - t[ca, 0]))
stats("   per K step", (t[ca, 1] - t[ca, 0]) / np.maximum(sa[ca], 1))
stats("   write-out", t[ca, 2] - t[ca, 1])
ob = (sb > 0) & (tb == 1)
start_b = np.where(ca, t[:, 2], t[:, 0])
stats("first part / whole tile: main", t[ob, 3] - start_b[ob])
stats("   per K step", (t[ob, 3] - start_b[ob]) / np.maximum(sb[ob], 1))
sp = ob & (s[:, 5] != 0)
stats("   split: write-out", t[sp, 4] - t[sp, 3])
stats("   split: count + inline fold", t[sp, 5] - t[sp, 4])
un = ob & (s[:, 5] == 0)
stats("   whole: epilogue", t[un, 5] - t[un, 3])
stats("after the K walk: waits + folds", t[:, 6] - t[:, 5])
stats("whole workgroup", t[:, 6] - t[:, 0])
