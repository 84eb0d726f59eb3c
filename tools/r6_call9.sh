#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/r6_call9; rm -rf $O; mkdir -p $O
export SWEEP_BRIEF=0 SWEEP_COLD=1
for f in "d9 lin" "thin lin" "t18 lin" "t16 lin" "d9 conv" "t9 conv 256x16" "thin conv 128x8 b9" "thin conv 256x16 b9"; do
  SWEEP_FILTER="$f" timeout 600 python3 tools/sweep_conv.py 2>&1 | grep -v "amdgpu.ids"
done | tee $O/sweep_all.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 python3 $R/bench.py --mode perceptual --no-cpu-baseline --steps 3 --warmup 1 --perceptual-fused-batch 45 2> $O/perc.err | tail -1 > $O/perc.json
python3 - $O/perc.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ms_per_step", "fused_micro_batch")})
PY
