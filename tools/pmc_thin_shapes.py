#!/usr/bin/env python3
"""HBM-side read traffic per launch of the thin K-heavy conv_gemm shapes of the distillation step, with and
without the weight-slab XCD mapping.

  run:    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o p -- python3 tools/pmc_thin_shapes.py run DIR/manifest.json
  parse:  python3 tools/pmc_thin_shapes.py parse DIR [DIR2 ...]     (one DIR per library build / option setting; prints one table)

`run` launches every shape REPS times on rotating weight copies (cold, as inside the pipeline) and writes the manifest the
parser needs to cut the profiler's dispatch list into shapes (one conv_gemm_kernel dispatch per launch, in launch order).
FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950: x2 (MI355X_MICROARCH.md)."""
import ctypes
import csv
import glob
import json
import os
import sys

SHAPES = [  # (tag, B, H, W, Cin, Cout, k)
    ("conv 32x2 b9  1024>1024", 9, 32, 2, 1024, 1024, 3),
    ("conv 32x2 b18 1024>1024", 18, 32, 2, 1024, 1024, 3),
    ("conv 64x4 b9  1024>1024", 9, 64, 4, 1024, 1024, 3),
    ("conv 64x4 b18 1024>1024", 18, 64, 4, 1024, 1024, 3),
    ("conv 64x4 b9  2048>1024", 9, 64, 4, 2048, 1024, 3),
    ("conv 32x2 b32 1024>1024", 32, 32, 2, 1024, 1024, 3),
    ("conv 64x4 b32 1024>1024", 32, 64, 4, 1024, 1024, 3),
    ("lin M2304 4096>1024", 9, 256, 1, 4096, 1024, 1),
    ("lin M4608 4096>1024", 18, 256, 1, 4096, 1024, 1),
]
REPS = 6


def run(manifest):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from consistencytta_amd import _native as N
    L = N.lib()
    out = []
    for tag, B, H, W, Cin, Cout, k in SHAPES:
        x = (torch.randn(B, H, W, Cin, device="cuda:0") * 0.5).to(torch.bfloat16)
        o = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device="cuda:0")
        K = k * k * Cin
        ncopy = max(2, (600 << 20) // (Cout * K * 2) + 1)
        ws = [(torch.randn(Cout, K, device="cuda:0") * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
        bias = torch.randn(Cout, device="cuda:0")
        d = N.ConvDesc()
        d.x0, d.c0 = x.data_ptr(), Cin
        d.batch, d.hi, d.wi, d.ho, d.wo = B, H, W, H, W
        d.kh, d.kw, d.stride_h, d.stride_w, d.dil_h, d.dil_w = k, k, 1, 1, 1, 1
        d.pad_h, d.pad_w = (k - 1) // 2, (k - 1) // 2
        d.k_pad, d.n, d.bias = K, Cout, bias.data_ptr()
        d.alpha, d.groups, d.out, d.ldc, d.tile = 1.0, 1, o.data_ptr(), Cout, 0
        for i in range(REPS):
            d.w = ws[i % ncopy].data_ptr()
            N.check(L.ctta_conv_gemm(ctypes.byref(d), N.stream_ptr()))
        torch.cuda.synchronize()
        M = B * H * W
        out.append({"tag": tag, "M": M, "N": Cout, "K": K, "launches": REPS,
                    "operand_MB": round((M * Cin * 2 + Cout * K * 2) / 1e6, 2), "weight_MB": round(Cout * K * 2 / 1e6, 2)})
        del x, o, ws
    json.dump(out, open(manifest, "w"), indent=1)


def parse(dirs):
    cols = []
    man = None
    for dname in dirs:
        man = json.load(open(os.path.join(dname, "manifest.json")))
        path = glob.glob(dname + "/**/*counter_collection.csv", recursive=True)[0]
        disp = {}
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == "FETCH_SIZE" and "conv_gemm_kernel" in row["Kernel_Name"]:
                disp[int(row["Dispatch_Id"])] = disp.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
        vals = [v for _, v in sorted(disp.items())]
        assert len(vals) == sum(m["launches"] for m in man), (len(vals), dname)
        per, i = [], 0
        for m in man:
            chunk = vals[i:i + m["launches"]][2:]          # the first launches see warm-up effects
            per.append(sum(chunk) / len(chunk) * 1024 * 2 / 1e6)
            i += m["launches"]
        cols.append((os.path.basename(dname.rstrip("/")), per))
    print("%-26s %7s %6s %7s %10s %10s  %s" % ("shape", "M", "N", "K", "operand MB", "weight MB",
                                               "  ".join("%s: read MB (x operand)" % c[0] for c in cols)))
    for j, m in enumerate(man):
        print("%-26s %7d %6d %7d %10.1f %10.1f  %s" % (m["tag"], m["M"], m["N"], m["K"], m["operand_MB"], m["weight_MB"],
              "  ".join("%8.1f (%4.2f)" % (c[1][j], c[1][j] / m["operand_MB"]) + " " * (len(c[0]) + 4) for c in cols)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        parse(sys.argv[2:])
