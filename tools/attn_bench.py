#!/usr/bin/env python3
"""Self-attention microbenchmark through the C ABI: executed TFLOP/s (2 * 2 * nq * nk * 64 per head) and max error vs an fp32
torch reference on the same bf16 operands.  (Until round 5 an environment knob selected the round-2 kernel; the library reads no environment any more.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def main():
    L = N.lib()
    shapes = [(32, 5, 4096, 51), (32, 10, 1024, 51), (32, 20, 256, 51), (9, 5, 4096, 51), (2, 3, 320, 51)]
    if os.environ.get("ATTN_SHAPES"):            # e.g. ATTN_SHAPES=0 for counter passes: one shape only
        shapes = [shapes[int(i)] for i in os.environ["ATTN_SHAPES"].split(",")]
    for (B, H, n, dh) in shapes:
        hp = H * 64
        g = torch.Generator().manual_seed(n)
        q = torch.zeros(B, n, hp)
        k = torch.zeros(B, n, hp)
        v = torch.zeros(B, n, hp)
        for t in (q, k, v):
            t.view(B, n, H, 64)[..., :dh] = torch.randn(B, n, H, dh, generator=g)
        qd, kd = q.to(torch.bfloat16).to(DEV), k.to(torch.bfloat16).to(DEV)
        vt_ld = (n + 7) // 8 * 8
        vt = torch.zeros(B, hp, vt_ld, dtype=torch.bfloat16, device=DEV)
        vt[:, :, :n] = v.to(torch.bfloat16).to(DEV).transpose(1, 2)
        out = torch.empty(B, n, hp, dtype=torch.bfloat16, device=DEV)
        scale = 1.0 / dh ** 0.5
        s = N.stream_ptr()

        def run():
            N.check(L.ctta_attention(N.ptr(qd), hp, N.ptr(kd), hp, n, N.ptr(vt), vt_ld, None, N.ptr(out), hp, B, H, n, n,
                                     scale, s))
        run()
        torch.cuda.synchronize()
        # reference on a slice (first 2 batches) in fp32
        nb = min(B, 2)
        qq = qd[:nb].float().view(nb, n, H, 64).transpose(1, 2)
        kk = kd[:nb].float().view(nb, n, H, 64).transpose(1, 2)
        vv = vt[:nb, :, :n].float().view(nb, H, 64, n).transpose(2, 3)
        ref = torch.softmax(qq @ kk.transpose(2, 3) * scale, -1) @ vv
        got = out[:nb].float().view(nb, n, H, 64).transpose(1, 2)
        err = float((got - ref).abs().max() / ref.abs().max())
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ts = []
        for _ in range(5):
            e[0].record()
            for _ in range(5):
                run()
            e[1].record()
            torch.cuda.synchronize()
            ts.append(e[0].elapsed_time(e[1]) / 5)
        ms = sorted(ts)[len(ts) // 2]
        fl = 4.0 * n * n * 64 * B * H
        print("B%d H%d n%d: %.3f ms  %.1f TF/s executed  (max err %.2e)" % (B, H, n, ms, fl / ms / 1e9, err), flush=True)


if __name__ == "__main__":
    main()
