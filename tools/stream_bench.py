#!/usr/bin/env python3
"""AdamW / EMA streaming microbenchmark through the C ABI at the distillation model's parameter count: milliseconds and
TB/s of algorithmic bytes (AdamW 28 B / parameter: p, g, m, v read, p, m, v written; two-shadow EMA 20 B: p, a, b read,
a, b written).
(Round 5: nontemporal loads / stores of the moments and shadows and workgroup caps 1024..8192 measured within +-4 % of the
shipped kernels -- 2.48-2.74 ms AdamW, 2.02-2.23 ms EMA at 523 M parameters -- and were not kept.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=5, inner=4):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(reps):
        ev[0].record()
        for _ in range(inner):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / inner)
    return sorted(ts)[len(ts) // 2]


def main():
    L = N.lib()
    n = int(os.environ.get("STREAM_N", str(523_000_000))) // 4 * 4
    p = torch.randn(n, device=DEV) * 0.02
    g = torch.randn(n, device=DEV) * 1e-3
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    a = p.clone()
    b = p.clone()
    s = N.stream_ptr()
    step = [0]

    def adamw():
        step[0] += 1
        N.check(L.ctta_adamw_step(N.ptr(p), N.ptr(g), N.ptr(m), N.ptr(v), n, 1e-5, 0.9, 0.999, 1e-8, 0.01, step[0], 1.0, s))

    def ema():
        N.check(L.ctta_ema_update2(N.ptr(p), N.ptr(a), 0.95, N.ptr(b), 0.999, n, s))

    def both():
        adamw()
        ema()
    adamw(); ema()
    torch.cuda.synchronize()
    t1, t2, t3 = timed(adamw), timed(ema), timed(both)
    print("n=%d  adamw %.3f ms (%.2f TB/s)  ema2 %.3f ms (%.2f TB/s)  adamw+ema2 %.3f ms"
          % (n, t1, 28.0 * n / t1 / 1e9, t2, 20.0 * n / t2 / 1e9, t3), flush=True)


if __name__ == "__main__":
    main()
