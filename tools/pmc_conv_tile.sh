set -u
R=$PWD; O=$R/gpurun_out/pmc29; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $R/tools/pmc_probe.py 0 29 3 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/p2 -o p -- python3 $R/tools/pmc_probe.py 0 29 3 > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p3 -o p -- python3 $R/tools/pmc_probe.py 0 29 3 > $O/p3.log 2>&1
for p in p1 p2 p3; do f=$(find $O/$p -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "conv_gemm" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in tot: print("%-34s %.4e  (%d rows)" % (k, tot[k], n[k]))
PY
done
tail -2 $O/p3.log
