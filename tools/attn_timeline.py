#!/usr/bin/env python3
"""Per-wave phase timeline of ONE self-attention forward launch (ctta_attention_debug_stamps: s_memtime at six points of
every key tile, per wave): where a tile's time goes -- waiting at the top (LDS-DMA landed + workgroup barrier), LDS-DMA
issue + the next tile's score MFMAs, softmax, P V -- and how the waves that share a SIMD interleave.
usage: attn_timeline.py [B H n]      default 32 5 4096"""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consistencytta_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def main():
    B, H, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 5, 4096)
    L = N.lib()
    hp = H * 64
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, n, hp, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(B, n, hp, generator=g).to(torch.bfloat16).to(DEV)
    vt = torch.randn(B, hp, n, generator=g).to(torch.bfloat16).to(DEV)
    out = torch.empty(B, n, hp, dtype=torch.bfloat16, device=DEV)
    s = N.stream_ptr()

    def run():
        N.check(L.ctta_attention(N.ptr(q), hp, N.ptr(k), hp, n, N.ptr(vt), n, None, N.ptr(out), hp, B, H, n, n, 0.125, s))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    plain_ms = e0.elapsed_time(e1)
    nwg = (n + 127) // 128 * B * H
    per = 4 * 64 * 6 + 8
    buf = torch.zeros(nwg * per, dtype=torch.int32, device=DEV)
    L.ctta_attention_debug_stamps(buf.data_ptr())
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    L.ctta_attention_debug_stamps(None)
    stamp_ms = e0.elapsed_time(e1)
    a = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    a = a.reshape(nwg, per)
    st = a[:, :1536].reshape(nwg, 4, 64, 6)
    hw = a[:, 1536:1540]
    xcc = a[:, 1540:1544]
    nt = min(64, n // 64)
    st = st[:, :, :nt]
    d = np.diff(st, axis=3) & 0xFFFFFFFF                       # [wg, wave, tile, 4]: wait, issue+scores, softmax, PV
    tile = ((st[:, :, 1:, 0] - st[:, :, :-1, 0]) & 0xFFFFFFFF)  # top to top
    names = ["wait at the top (vmcnt + barrier)", "LDS-DMA issue (4 instructions)", "next tile's scores (12 ds_read, 16 MFMA)", "softmax", "P V"]
    print("B %d, %d heads, %d tokens: %.3f ms plain, %.3f ms with stamps; %d workgroups; s_memtime ticks" % (B, H, n, plain_ms, stamp_ms, nwg))
    mid = d[:, :, 4:nt - 4]
    tot = float(tile[:, :, 4:nt - 4].mean())
    for i, nm in enumerate(names):
        v = mid[..., i].astype(np.float64)
        print("  %-44s mean %7.0f  median %7.0f  p90 %7.0f   (%4.1f %% of a tile)" % (nm, v.mean(), np.median(v), np.percentile(v, 90), 100.0 * v.mean() / tot))
    print("  %-44s mean %7.0f  median %7.0f" % ("tile (top to top)", tot, float(np.median(tile[:, :, 4:nt - 4]))))
    span = ((st[:, :, nt - 1, 5] - st[:, :, 0, 0]) & 0xFFFFFFFF).astype(np.float64)
    print("  wave lifetime inside the loop: mean %.0f ticks = %.0f per tile" % (span.mean(), span.mean() / nt))
    # waves sharing a SIMD: group by (xcc, se, cu, simd) and look at overlap of their [first top, last end] intervals
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    key = ((xcc & 15) << 12) | (se << 8) | (cu << 4) | simd
    groups = defaultdict(list)
    for w in range(nwg):
        for j in range(4):
            groups[int(key[w, j])].append((int(st[w, j, 0, 0]), int(st[w, j, nt - 1, 5]), w, j))
    print("  distinct (xcc, se, cu, simd): %d" % len(groups))
    # concurrency on one SIMD: for a sample of SIMDs, the average number of resident stamped waves over time
    conc = []
    for kk, lst in list(groups.items())[:64]:
        ev = []
        for (b0, e, w, j) in lst:
            if e < b0:
                continue
            ev += [(b0, 1), (e, -1)]
        ev.sort()
        cur, last, area, busy = 0, None, 0, 0
        for tt, dd in ev:
            if last is not None and cur > 0:
                area += cur * (tt - last)
                busy += tt - last
            cur += dd
            last = tt
        if busy:
            conc.append(area / busy)
    print("  waves resident per SIMD while any is (sample of 64 SIMDs): %.2f" % (sum(conc) / max(len(conc), 1)))
    # one SIMD's interleaving: print the phases of its waves over 3 tiles in the middle
    kk, lst = max(groups.items(), key=lambda kv: len(kv[1]))
    lst.sort()
    sel = lst[len(lst) // 2: len(lst) // 2 + 3]
    t_ref = None
    print("  three waves that follow each other on one SIMD (key %#x), tiles 20..23: phase start times relative to the first" % kk)
    for (b0, e, w, j) in sel:
        row = st[w, j, 20:24]
        if t_ref is None:
            t_ref = int(row[0, 0])
        print("    wg %5d wave %d: " % (w, j) + "  ".join("[" + " ".join("%6d" % ((int(x) - t_ref) & 0xFFFFFFFF if ((int(x) - t_ref) & 0xFFFFFFFF) < 1 << 31 else -(((t_ref - int(x))) & 0xFFFFFFFF)) for x in r) + "]" for r in row))


if __name__ == "__main__":
    main()
