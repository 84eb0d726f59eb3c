#!/bin/bash
# round 5, call 2: thin K-heavy shapes -- ring depth x split-K target
set -u
R=$PWD
O=$R/gpurun_out/r5c2
rm -rf $O; mkdir -p $O
for tgt in 256 512 768 1024; do
  echo "== CTTA_SPLITK_TARGET=$tgt CTTA_SPLITK_MAX=16"
  CTTA_SPLITK_TARGET=$tgt CTTA_SPLITK_MAX=16 SWEEP_FILTER="thin conv" SWEEP_COLD=1 SWEEP_VARIANTS=22,27,37,17,38,40,24,39 python3 tools/sweep_conv.py
  CTTA_SPLITK_TARGET=$tgt CTTA_SPLITK_MAX=16 SWEEP_FILTER="thin lin M2304 4096" SWEEP_COLD=1 SWEEP_VARIANTS=22,27,37,17,38,40,24,39 python3 tools/sweep_conv.py
done 2>&1 | grep -v "amdgpu.ids" | tee $O/sweep.txt
