#!/bin/bash
set -u
O=$PWD/gpurun_out/r5c13; rm -rf $O; mkdir -p $O
for flt in "M131072" "u32 M32768" "u32 M8192" "t16 lin"; do
  SWEEP_FILTER="$flt" SWEEP_COLD=2 python3 tools/sweep_conv.py
done 2>&1 | grep -v amdgpu.ids | tee $O/sweep_linears.txt
