#!/usr/bin/env python3
"""Per-shape HBM read traffic of conv_gemm launches: joins a rocprofv3 --pmc FETCH_SIZE pass of `bench.py --mode gen`
with the launch records of `--profile-csv` by dispatch order (the pipeline issues the same launch sequence every step).
usage: pmc_by_shape.py <pmc_dir> <launch_csv> [steps_in_pmc_run=9]"""
import csv
import glob
import sys
from collections import OrderedDict

pmc_dir, launch_csv = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 9
path = glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True)[0]
disp = OrderedDict()
for row in csv.DictReader(open(path)):
    if row["Counter_Name"] != "FETCH_SIZE":
        continue
    n = row["Kernel_Name"]
    if "conv_gemm_kernel" in n or "conv1d_halo_kernel" in n:
        disp.setdefault(int(row["Dispatch_Id"]), [n, 0.0])[1] += float(row["Counter_Value"])
vals = [v for _, v in sorted(disp.items())]
recs = [r.strip().split(",") for r in open(launch_csv)]
recs = [r for r in recs if r[0] == "0"]
per_step = len(vals) // steps
nrec = len(recs) // 2                      # the csv holds two profiled steps
print("conv_gemm dispatches per step: pmc %d, launch csv %d" % (per_step, nrec))
assert per_step == nrec, "sequences do not line up"
step = vals[3 * per_step:4 * per_step]     # a steady-state step
agg = OrderedDict()
for (name, kib), r in zip(step, recs[:nrec]):
    var, m, n, k, g = int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])
    a = agg.setdefault((var, m, n, k, g), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += kib * 1024 * 2 / 1e6           # MB, gfx950 x2 correction for 16 B/lane reads
    a[2] += float(r[6])
print("var         M      N      K    G  cnt  read MB/launch  in+w MB (algorithmic)  ratio   ms/launch  read GB/s")
tot = 0.0
for (var, m, n, k, g), (cnt, mb, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    tot += mb
    # algorithmic reads: the input tensor once (M rows x K/taps channels is unknown here: bound it by M*min(K,N*?)...)
    print("%3d  %9d %6d %6d %4d %4d  %12.1f  %8.3f %10.1f" % (var, m, n, k, g, cnt, mb / cnt, ms / cnt, mb / 1e3 / (ms * 1e-3)))
print("total read of listed shapes: %.1f GB; all: %.1f GB" % (tot / 1e3, sum(v[1] for v in agg.values()) / 1e3))
