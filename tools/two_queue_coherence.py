#!/usr/bin/env python3
"""Two hardware queues, one chain of dependent kernels: does the consumer see the producer's data?

    python tools/two_queue_coherence.py {none|conv|lin|torch|conv_torchmain}      (GPU box)

Main stream: y = W2 silu(W x + b) + b as two ctta_linear_f32 launches on fresh random x every iteration, checked against
torch.  Side stream (non-blocking), concurrently: nothing / three big-tile ctta_conv_gemm launches / fifty ctta_linear_f32
launches / twenty torch matmuls.  Measured on MI355X, ROCm 7.0 runtime of the PyTorch wheel: `conv` 40 of 40 iterations
wrong (stale reads: the rerun without concurrency is right), `none` / `lin` / `torch` 0 of 40, GPU_MAX_HW_QUEUES=1 0 of 40,
agent-scope atomic loads in the consumer 0 of 40 (round-2 build WITH the SLP vectoriser).  The cause turned out to be an
instruction form, not coherence: tools/pk_hazard.py, LABNOTES.md 5; built with -fno-slp-vectorize this script reports 0 of 40
in every mode.  MAIN2=1 runs the chain on an explicit stream instead of the null stream, TILE=n picks the conv variant."""
import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from consistencytta_amd import _native as N
DEV = torch.device("cuda:0")
mode = sys.argv[1]
L = N.lib()
M, K, Nn = 18, 256, 1024
W = torch.randn(Nn, K, device=DEV) * 0.05; b = torch.randn(Nn, device=DEV); W2 = torch.randn(Nn, Nn, device=DEV) * 0.03
def chain(x):
    h = torch.empty(M, Nn, device=DEV); y = torch.empty(M, Nn, device=DEV)
    if mode.endswith("torcheltmain"):      # torch elementwise chain through fresh buffers (returned as (h, y) lookalikes)
        a1 = x * 2.0; a2 = a1 + 1.0; a3 = a2.sin(); a4 = a3 * 3.0; a5 = a4.cos(); got = a5 + x
        refc = ((x * 2.0 + 1.0).sin() * 3.0).cos() + x if False else None
        return got, got
    if mode.endswith("torchmain"):
        h = torch.nn.functional.silu(x @ W.t() + b); y = h @ W2.t() + b
        return h, y
    N.check(L.ctta_linear_f32(N.ptr(x), N.ptr(W), N.ptr(b), N.ptr(h), M, Nn, K, 0, 1, N.stream_ptr()))
    N.check(L.ctta_linear_f32(N.ptr(h), N.ptr(W2), N.ptr(b), N.ptr(y), M, Nn, Nn, 0, 0, N.stream_ptr()))
    return h, y
side = torch.cuda.Stream()
# side-stream load generators
C = 256
xs = (torch.randn(32, 1, 20484, C, device=DEV) * 0.5).to(torch.bfloat16); outs = torch.empty_like(xs)
ws = (torch.randn(C, 11 * C, device=DEV) * 0.05).to(torch.bfloat16); bs = torch.randn(C, device=DEV)
d = N.ConvDesc()
d.x0, d.c0 = xs.data_ptr(), C
d.batch, d.hi, d.wi, d.ho, d.wo = 32, 1, 20484, 1, 20484
d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = 1, 11, 1, 1, 1, 1
d.pad_h, d.pad_w = 0, 5
d.w, d.k_pad, d.n, d.bias = ws.data_ptr(), 11 * C, C, bs.data_ptr()
d.alpha, d.groups, d.out, d.ldc, d.tile = 1.0, 1, outs.data_ptr(), C, int(os.environ.get('TILE', '0'))
xa = torch.randn(64, 1024, device=DEV); Wa = torch.randn(4096, 1024, device=DEV); ba = torch.randn(4096, device=DEV); ya = torch.empty(64, 4096, device=DEV)
BIG = torch.randn(16384, 16384, device=DEV) if mode.startswith('bigtorch') else None; BIG2 = torch.randn(16384, 16384, device=DEV) if mode.startswith('bigtorch') else None
if mode.startswith('gn'):
    GX = torch.randn(32, 65536, 128, device=DEV).to(torch.bfloat16); GY = torch.empty_like(GX); GG = torch.ones(128, device=DEV); GB = torch.zeros(128, device=DEV)
    GS = torch.empty(int(L.ctta_groupnorm_scratch_floats(32, 65536, 128, 32)), device=DEV)
bad = 0
for i in range(40):
    x = torch.randn(M, K, device=DEV)          # new values every iteration: a stale read cannot hide
    torch.cuda.synchronize()
    ref_h = torch.nn.functional.silu(x @ W.t() + b); ref_y = ref_h @ W2.t() + b
    if mode.endswith('torcheltmain'): ref_h = ref_y = ((x * 2.0 + 1.0).sin() * 3.0).cos() + x
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        if mode.startswith("conv"):
            for _ in range(3): N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
        elif mode.startswith("lin"):
            for _ in range(50): N.check(L.ctta_linear_f32(N.ptr(xa), N.ptr(Wa), N.ptr(ba), N.ptr(ya), 64, 4096, 1024, 0, 1, N.stream_ptr()))
        elif mode.startswith("torch"):
            for _ in range(20): _ = Wa @ Wa.t()
        elif mode.startswith("gn"):          # a plain streaming kernel of this library as the co-runner
            N.check(L.ctta_groupnorm(N.ptr(GX), N.ptr(GY), 32, 65536, 128, 32, N.ptr(GG), N.ptr(GB), 1e-5, 1, N.ptr(GS), N.stream_ptr()))
        elif mode.startswith("bigtorch"):
            _ = BIG @ BIG2                                  # one ~10 ms kernel on every CU
    if os.environ.get('MAIN2') == '1':
        main2 = globals().setdefault('_m2', torch.cuda.Stream())
        main2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main2):
            h, y = chain(x)
    else:
        h, y = chain(x)
    torch.cuda.synchronize()
    eh, ey = float((h - ref_h).abs().max()), float((y - ref_y).abs().max())
    h2, y2 = chain(x); torch.cuda.synchronize()
    if float((h2 - ref_h).abs().max()) > 1e-3 or float((y2 - ref_y).abs().max()) > 1e-3: print('  persistent: rerun without concurrency also wrong')
    if eh > 1e-3 or ey > 1e-3:
        bad += 1
        if bad <= 3: print("iter", i, "h err %.3e y err %.3e" % (eh, ey))
print(mode, "bad iterations:", bad, "of 40")
