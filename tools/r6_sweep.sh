#!/bin/bash
# Round 6: parity of the new tile kinds, then the sweeps that decide their rules (gpurun --timeout 1500 -- 'bash tools/r6_sweep.sh')
set -u
R=$PWD; O=$R/gpurun_out/r6_sweep; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "all_tiles or mf32_and_streamk or streamk" 2>&1 | tail -15 | tee $O/tests.txt
export SWEEP_BRIEF=0
for f in "vae" "hifi L5121" "hifi L20484" "hifi L40968" "unet "; do
  SWEEP_FILTER="$f" SWEEP_VARIANTS=29,41,31,42,36,43 timeout 300 python3 tools/sweep_conv.py 2>&1 | grep -v "amdgpu.ids"
done | tee $O/sweep_mf32.txt
for f in "thin conv" "thin lin M2304 4096" "t16 conv" "t18 conv" "t9 conv" "u32 conv" "lin M8192 4096" "t16 lin M4096 4096" "t18 lin M4608 4096"; do
  SWEEP_COLD=1 SWEEP_FILTER="$f" SWEEP_VARIANTS=17,22,27,29,41,44,45,46,47,48 timeout 300 python3 tools/sweep_conv.py 2>&1 | grep -v "amdgpu.ids"
done | tee $O/sweep_sk.txt
