#!/bin/bash
# PMC passes over one short-K linear (sweep_conv.SHAPES index 21: M=131072, N=256, K=256), variant $1 (default 18).
set -u
V=${1:-18}
R=$PWD; O=$R/gpurun_out/pmc_shortk; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -o p -- python3 $R/tools/pmc_probe.py 21 $V 5 > $O/$1.log 2>&1; }
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
run p2 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM"
run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE"
run p5 "TCC_HIT_sum TCC_MISS_sum"
for p in p1 p2 p3 p4 p5; do f=$(find $O/$p -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_gemm" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in tot: print("%-34s %.4e  (%d rows)" % (k, tot[k], n[k]))
PY
done
f=$(find $O/p1 -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_gemm" in r["Kernel_Name"]:
        print("duration_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"), "lds", r.get("LDS_Block_Size"), "vgpr", r.get("VGPR_Count"))
PY
