#!/bin/bash
# Same-box A/B between the library in tmp_head/ (built from another revision or with other flags) and the in-tree one:
#   gpurun --timeout 900 -- 'bash tools/lib_ab.sh "<command>" [rounds]'      e.g. "SWEEP_FILTER=vae SWEEP_VARIANTS=29 python3 tools/sweep_conv.py"
set -u
R=$PWD
cmd=$1; rounds=${2:-2}
cp consistencytta_amd/libctta_hip.so /tmp/new.so
for r in $(seq $rounds); do
  for v in head new; do
    echo "== $v"
    if [ $v = head ]; then cp tmp_head/libctta_hip.so consistencytta_amd/libctta_hip.so; else cp /tmp/new.so consistencytta_amd/libctta_hip.so; fi
    bash -c "$cmd" 2>&1 | grep -v "amdgpu.ids\|^variants"
  done
done
cp /tmp/new.so consistencytta_amd/libctta_hip.so
