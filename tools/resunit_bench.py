#!/usr/bin/env python3
"""HiFi-GAN ResBlock unit (ctta_resunit_conv1d, csrc/resunit.hip) against the two conv_gemm launches it replaces, at the
vocoder's stage shapes (B = 32): ms per unit and algorithmic TFLOP/s (2 x 2 M C kC).  RU_SHAPES="512:5121,256:20484" and
RU_TAPS="3:1,7:3,11:5" (k:dilation) select the cases; the fused column is timed twice (first reading = cold clocks)."""
import ctypes, math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from consistencytta_amd import _native as N
from gpu_util import conv_desc, pack_conv_weight
DEV = "cuda:0"
L_ = N.lib()
def bench(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(5):
        e[0].record()
        for _ in range(5): fn()
        e[1].record(); torch.cuda.synchronize()
        ts.append(e[0].elapsed_time(e[1]) / 5)
    return sorted(ts)[2]
SHAPES = [tuple(int(v) for v in a.split(":")) for a in os.environ.get("RU_SHAPES", "512:5121,256:20484,128:40968,64:81936,32:163872").split(",")]
TAPS = [tuple(int(v) for v in a.split(":")) for a in os.environ.get("RU_TAPS", "3:1,7:3,11:5").split(",")]
for C, Lq in SHAPES:
  for k, d in TAPS:
    B = 32
    if not L_.ctta_resunit_supported(C, k, d):
        print("C %d k %d d %d: not served by the unit kernel" % (C, k, d)); continue
    g = torch.Generator().manual_seed(0)
    xa = torch.randn(B, Lq, C, generator=g).to(torch.bfloat16).to(DEV)
    frags = []
    for i in range(2):
        w = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
        wp, k_pad = pack_conv_weight(w[:, :, None, :])
        f = torch.empty(C * k * C, dtype=torch.bfloat16, device=DEV)
        N.check(L_.ctta_frag_pack(N.ptr(wp), C, k_pad, k * C, N.ptr(f), N.stream_ptr()))
        frags.append((f, wp, k_pad))
    b1d, b2d = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    out = torch.empty_like(xa); t1 = torch.empty_like(xa); t2 = torch.empty_like(xa); o2 = torch.empty_like(xa)
    st = N.stream_ptr()
    def fused():
        N.check(L_.ctta_resunit_conv1d(N.ptr(xa), B, Lq, C, k, d, N.ptr(frags[0][0]), N.ptr(b1d), N.ptr(frags[1][0]), N.ptr(b2d), 0.1, N.ptr(out), 0, 1.0, 0.0, st))
    d1 = conv_desc(x0=xa, c0=C, batch=B, hi=1, wi=Lq, ho=1, wo=Lq, kh=1, kw=k, pad_w=(k * d - d) // 2, dil_w=d, w=frags[0][1], k_pad=frags[0][2], n=C, bias=b1d, out_act=3, out_slope=0.1, out=t1, ldc=C)
    d2 = conv_desc(x0=t1, c0=C, batch=B, hi=1, wi=Lq, ho=1, wo=Lq, kh=1, kw=k, pad_w=(k - 1) // 2, w=frags[1][1], k_pad=frags[1][2], n=C, bias=b2d, res=xa, res_ld=C, out=t2, ldc=C, out2=o2, out2_slope=0.1)
    def two():
        N.check(L_.ctta_conv_gemm(ctypes.byref(d1), st)); N.check(L_.ctta_conv_gemm(ctypes.byref(d2), st))
    fl = 2 * 2.0 * B * Lq * C * C * k
    tf, tt = bench(fused), bench(two)
    tf2 = bench(fused)
    print("C %d k %d d %d: fused %.3f / %.3f ms (%.0f TF/s)   two conv_gemm %.3f ms (%.0f TF/s)" % (C, k, d, tf, tf2, fl / tf2 / 1e9, tt, fl / tt / 1e9), flush=True)
