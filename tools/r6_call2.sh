#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/r6_call2; rm -rf $O; mkdir -p $O
timeout 120 python3 tools/r6_sk_debug.py 2>&1 | grep -v amdgpu.ids | tee $O/sk_debug.txt
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_bwd_ops_gpu.py tests/test_train_gpu.py -q -m gpu -k "all_tiles or mf32_and_streamk or streamk or optimizer_tail" 2>&1 | tail -25 | tee $O/tests.txt
export SWEEP_BRIEF=0
for g in 0 128 192; do
  echo "== streamk_grid $g"
  for f in "thin conv 64x4 b18" "thin conv 64x4 b9" "t16 conv 64x4 2048" "t18 conv 128x8 1024>512" "thin conv 128x8 b18" "t9 conv 256x16" "u32 conv 64x4" "t18 conv 256x16 512"; do
    CTTA_OPT_STREAMK_GRID=$g SWEEP_COLD=1 SWEEP_FILTER="$f" SWEEP_VARIANTS=17,27,29,44,45 timeout 300 python3 tools/sweep_conv.py 2>&1 | grep -v "amdgpu.ids\|^variants"
  done
done | tee $O/sweep_sk.txt
