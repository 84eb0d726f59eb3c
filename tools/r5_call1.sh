#!/bin/bash
# round 5, call 1: parity of the touched paths + A/B of the weight-slab XCD mapping on the thin shapes
set -u
R=$PWD
O=$R/gpurun_out/r5c1
rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_bwd_ops_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/ops.txt
timeout 1200 python3 -m pytest tests/test_train_gpu.py -x -q -m gpu -k "pipelined or full_batch or light_widths" 2>&1 | tail -8 | tee $O/train.txt
for v in 0 1; do
  echo "== CTTA_XCD_SLAB=$v"
  CTTA_XCD_SLAB=$v SWEEP_FILTER=thin SWEEP_COLD=1 SWEEP_BRIEF=1 SWEEP_VARIANTS=22 python3 tools/sweep_conv.py
  CTTA_XCD_SLAB=$v SWEEP_FILTER="u32 conv" SWEEP_COLD=1 SWEEP_BRIEF=1 SWEEP_VARIANTS=22 python3 tools/sweep_conv.py
  CTTA_XCD_SLAB=$v SWEEP_FILTER="t16 conv" SWEEP_COLD=1 SWEEP_BRIEF=1 SWEEP_VARIANTS=22 python3 tools/sweep_conv.py
done 2>&1 | grep -v "^variants" | tee $O/sweep.txt
