#!/usr/bin/env python3
"""U-Net forward only (light config, random weights) for profiling: `rocprofv3 --kernel-trace --stats -- python3
tools/prof_unet.py --batch 32 --guided 1 --iters 5`.  Prints the eager ms per forward measured with HIP events."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import modules, spec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--guided", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--text-len", type=int, default=32)
    ap.add_argument("--profile-csv", default=None, help="one line per MFMA launch of ONE forward (in-library HIP-event profiler)")
    a = ap.parse_args()
    dev = "cuda:0"
    cls = modules.UNet2DConditionGuidedModel if a.guided else modules.UNet2DConditionModel
    m = cls.from_config(spec.LIGHT_UNET_CONFIG).to(dev)
    m.init_random_(seed=1)
    m.eval().requires_grad_(False)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(a.batch, 8, 256, 16, generator=g).to(dev)
    enc = (torch.randn(a.batch, a.text_len, 1024, generator=g) * 0.25).to(dev)
    mask = torch.ones(a.batch, a.text_len, dtype=torch.bool, device=dev)
    t = torch.full((a.batch,), 999.0, device=dev)
    kw = dict(encoder_hidden_states=enc, encoder_attention_mask=mask)

    def fwd():
        if a.guided:
            return m(x, t, guidance=4.0, **kw).sample
        return m(x, t, **kw).sample
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fwd()
    e1.record()
    torch.cuda.synchronize()
    print("unet batch %d: %.3f ms per forward" % (a.batch, e0.elapsed_time(e1) / a.iters))
    if a.profile_csv:
        import ctypes
        from consistencytta_amd import _native as N
        L_ = N.lib()
        L_.ctta_prof_enable(1)
        fwd()
        torch.cuda.synchronize()
        L_.ctta_prof_enable(0)
        ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        N.check(L_.ctta_prof_collect(-1, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt), a.profile_csv.encode()))
        print("profiled forward: %.3f ms in %d MFMA launches, %.1f TF/s executed" % (ms.value, cnt.value, fl.value / ms.value / 1e9))


if __name__ == "__main__":
    main()
