#!/bin/bash
set -u
O=$PWD/gpurun_out/r5c4; rm -rf $O; mkdir -p $O
for args in "9 32 2 1024 1024 3 0" "9 32 2 1024 1024 3 27" "18 32 2 1024 1024 3 0" "18 32 2 1024 1024 3 17" "9 64 4 1024 1024 3 0" "18 64 4 1024 1024 3 0" "18 64 4 1024 1024 3 17"; do
  python3 tools/thin_timeline.py $args 1
done 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
