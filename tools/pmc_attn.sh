#!/bin/bash
# Counter passes over the self-attention forward kernel alone (tools/attn_bench.py, first shape: B 32, 5 heads, 4096 tokens).
# gpurun --timeout 900 -- 'bash tools/pmc_attn.sh'        
set -u
R=$PWD; O=$R/gpurun_out/pmc_attn; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export ATTN_SHAPES=0
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/tools/attn_bench.py > $O/p$i.log 2>&1
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "attention_plain2" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in tot: print("%-34s %.4e per launch (%d launches)" % (k, tot[k] / max(n[k], 1), n[k]))
PY
  tail -1 $O/p$i.log
  rm -rf $O/p$i
done 2>&1 | tee $O/pmc_attn.txt
