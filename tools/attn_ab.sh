#!/bin/bash
# A/B of the self-attention forward kernel forms on the GPU box: tools/attn_bench.py per CTTA_ATTN_V2 value, then the
# attention parity tests.  gpurun --timeout 900 -- 'bash tools/attn_ab.sh'
set -u
R=$PWD
O=$R/gpurun_out/attn_ab; rm -rf $O; mkdir -p $O
for v in ${ATTN_AB_VALUES:-0 1 0 1}; do
  echo "== CTTA_ATTN_V2=$v"
  CTTA_ATTN_V2=$v python3 $R/tools/attn_bench.py
done 2>&1 | grep -v "^$" | tee $O/attn_ab.txt
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "attention" 2>&1 | grep -E "passed|failed|Error" | tee $O/tests.txt
