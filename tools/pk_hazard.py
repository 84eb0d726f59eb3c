#!/usr/bin/env python3
"""v_pk_fma_f32 beside another kernel's MFMA waves: builds tools/pk_hazard.hip on the box and runs the synthetic victim
(packed vs scalar fma on identical inputs) alone and beside ctta_conv_gemm in a second stream."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N
so = "/tmp/libpk_hazard.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "pk_hazard.hip"), "-o", so], check=True)
L = N.lib()
PK = ctypes.CDLL(so)
PK.pk_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
DEV = torch.device("cuda:0")
C = 256
xs = (torch.randn(32, 1, 20484, C, device=DEV) * 0.5).to(torch.bfloat16); outs = torch.empty_like(xs)
ws = (torch.randn(C, 11 * C, device=DEV) * 0.05).to(torch.bfloat16); bs = torch.randn(C, device=DEV)
d = N.ConvDesc()
d.x0, d.c0 = xs.data_ptr(), C
d.batch, d.hi, d.wi, d.ho, d.wo = 32, 1, 20484, 1, 20484
d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = 1, 11, 1, 1, 1, 1
d.pad_h, d.pad_w = 0, 5
d.w, d.k_pad, d.n, d.bias = ws.data_ptr(), 11 * C, C, bs.data_ptr()
d.alpha, d.groups, d.out, d.ldc, d.tile = 1.0, 1, outs.data_ptr(), C, int(os.environ.get("TILE", "0"))
inp = torch.tensor([0.999, 0.001, 0.5, -0.25], device=DEV)
blocks, iters = 1024, 2000
n = blocks * 256 * 2
side = torch.cuda.Stream()
PK.co_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
co_out = torch.empty(2048 * 256, device=DEV); co_a = torch.randn(64 << 20, device=DEV); co_b = torch.empty_like(co_a)
MODES = [(m, v) for v in (0, 1, 2) for m in ("alone", "beside conv_gemm")] + [(m, 2) for m in ("beside MFMA-only kernel", "beside HBM-copy kernel", "beside fp32-VALU kernel")] + [("beside MFMA-only kernel", 0), ("beside MFMA-only kernel", 1)]
for mode, variant in MODES:
    bad_pk = bad_sc = 0
    big = torch.randn(8192, 8192, device=DEV) if "torch" in mode else None
    for it in range(20):
        o_pk = torch.zeros(n, device=DEV); o_sc = torch.zeros(n, device=DEV)
        torch.cuda.synchronize()
        if mode != "alone":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if "conv" in mode:
                    for _ in range(3): N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
                elif "MFMA-only" in mode:
                    PK.co_launch(0, co_out.data_ptr(), None, 0, 40000, torch.cuda.current_stream().cuda_stream)
                elif "HBM-copy" in mode:
                    for _ in range(3): PK.co_launch(1, co_a.data_ptr(), co_b.data_ptr(), co_a.numel() // 4, 0, torch.cuda.current_stream().cuda_stream)
                elif "VALU" in mode:
                    PK.co_launch(2, co_out.data_ptr(), None, 0, 400000, torch.cuda.current_stream().cuda_stream)
                else:
                    _ = big @ big
        PK.pk_launch(inp.data_ptr(), o_pk.data_ptr(), o_sc.data_ptr(), blocks, iters, variant, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if it == 0 and mode == "alone" and variant == 0: ref = o_sc.clone()
        bad_pk += int((o_pk != ref).sum()); bad_sc += int((o_sc != ref).sum())
    print("variant %d %-26s wrong packed results %d, wrong scalar results %d (of %d per launch x 20)" % (variant, mode, bad_pk, bad_sc, n))
