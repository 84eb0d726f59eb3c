#!/bin/bash
# round 5, call 3: how much of the thin shapes' time is memory latency?  hot (one weight copy: MALL / L2 resident) vs cold
set -u
R=$PWD
O=$R/gpurun_out/r5c3
rm -rf $O; mkdir -p $O
for cold in 0 1; do for slab in 0 1; do
  echo "== SWEEP_COLD=$cold CTTA_XCD_SLAB=$slab"
  CTTA_XCD_SLAB=$slab SWEEP_FILTER="thin conv" SWEEP_COLD=$cold SWEEP_VARIANTS=22,27,17,24 python3 tools/sweep_conv.py
done; done 2>&1 | grep -v "amdgpu.ids" | tee $O/sweep.txt
