#!/usr/bin/env python3
"""Round 6 debug aid: where a stream-K tile's output differs from the one-tile-per-workgroup result."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from consistencytta_amd import _native as N
from gpu_util import DEV, bf16_round, conv_desc, det, nhwc_bf16, pack_conv_weight, run_conv


def case(tile, B, Cin, H, W, Cout, grid=0):
    N.set_option("streamk_grid", grid)
    x = bf16_round(det("d.x", (B, Cin, H, W), 1))
    w = bf16_round(det("d.w", (Cout, Cin, 3, 3), 2) * (1.0 / math.sqrt(Cin * 9)))
    wp, k_pad = pack_conv_weight(w)
    xa = nhwc_bf16(x)
    outs = []
    for t in (17, tile):
        out = torch.zeros(B, H, W, Cout, dtype=torch.float32, device=DEV)
        run_conv(conv_desc(x0=xa, c0=Cin, batch=B, hi=H, wi=W, ho=H, wo=W, kh=3, kw=3, pad_h=1, pad_w=1, w=wp, k_pad=k_pad, n=Cout,
                           out=out, ldc=Cout, out_f32=1, tile=t))
        outs.append(out.reshape(-1, Cout).cpu())
    d = (outs[0] - outs[1]).abs()
    bad = (d > 1e-2 * outs[0].abs().max()).float()
    print("tile %d B=%d Cin=%d HxW=%dx%d Cout=%d grid=%d: M=%d max abs diff %.4g (ref max %.4g); bad elements %d" % (
        tile, B, Cin, H, W, Cout, grid, d.shape[0], float(d.max()), float(outs[0].abs().max()), int(bad.sum())))
    if bad.sum() > 0:
        rows = bad.sum(1).nonzero().flatten().tolist()
        cols = bad.sum(0).nonzero().flatten().tolist()
        print("   bad rows: %d in [%d, %d]  bad cols: %d in [%d, %d]" % (len(rows), rows[0], rows[-1], len(cols), cols[0], cols[-1]))
        print("   bad 16-row blocks:", sorted(set(r // 16 for r in rows))[:48])
        r, q = outs[1][rows[0]], outs[0][rows[0]]
        print("   row %d: got/ref at the first bad cols:" % rows[0], [(c, round(float(r[c]), 3), round(float(q[c]), 3)) for c in cols[:6]])


for tile in (43, 41):
    case(tile, 1, 64, 8, 16, 96)      # M = 128
    case(tile, 2, 64, 24, 16, 96)     # the failing test case
    case(tile, 2, 64, 24, 16, 96, grid=2)
    case(tile, 2, 64, 24, 16, 96, grid=1)
    case(tile, 4, 256, 16, 16, 256, grid=7)
