#!/bin/bash
# round 5: MODE 4 tiles (weights straight into MFMA-layout registers): parity, then the thin / mid shapes against the LDS-DMA tiles
set -u
R=$PWD
O=$R/gpurun_out/r5c10; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "all_tiles or geometries or split_k" 2>&1 | tail -6 | tee $O/ops.txt
for flt in "thin conv" "thin lin" "t16 conv" "u32 conv"; do
  SWEEP_FILTER="$flt" SWEEP_COLD=1 SWEEP_VARIANTS=22,27,17,41,42,43 python3 tools/sweep_conv.py
done 2>&1 | grep -v "amdgpu.ids" | tee $O/sweep.txt
for args in "18 64 4 1024 1024 3 22" "18 64 4 1024 1024 3 41" "18 64 4 1024 1024 3 42" "18 64 4 1024 1024 3 43" "9 128 8 512 512 3 22" "9 128 8 512 512 3 41"; do
  python3 tools/thin_timeline.py $args 1 2>&1 | grep -v amdgpu.ids | head -6
done | tee $O/timeline.txt
