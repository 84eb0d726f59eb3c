#!/usr/bin/env python3
"""GroupNorm+SiLU rate against the tensor size: does the apply pass re-read x from the Infinity Cache when the
tensor is small enough?  (decides whether large GroupNorms should run in sub-batches)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

dev = torch.device("cuda", 0)
L = N.lib()
for HW, C in ((65536, 128), (16384, 256), (4096, 512)):
    for B in (1, 2, 4, 8, 16, 32):
        x = torch.randn(B, HW, C, device=dev).to(torch.bfloat16)
        y = torch.empty_like(x)
        ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        scr = torch.empty(int(L.ctta_groupnorm_scratch_floats(B, HW, C, 32)), dtype=torch.float32, device=dev)

        def fn():
            N.check(L.ctta_groupnorm(N.ptr(x), N.ptr(y), B, HW, C, 32, N.ptr(ga), N.ptr(be), 1e-5, 1, N.ptr(scr), N.stream_ptr()))
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        mb = x.numel() * 2 / 1e6
        print("HW=%6d C=%4d B=%2d  x=%7.1f MB  %.3f ms  %.0f GB/s (2 passes)  %.3f us/MB" % (HW, C, B, mb, ms, 2 * mb / ms, ms * 1e3 / mb))
