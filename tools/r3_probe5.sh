#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p5
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
python3 tools/gn_fuse_ab.py > $O/ab.txt 2>&1
CTTA_GN_APPLY_FUSED=0 python3 tools/gn_fuse_ab.py > $O/ab_nofuseapply.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for cfg in "0 1" "1 0" "1 1" "0 1" "1 1"; do
  set -- $cfg
  CTTA_GN_FUSE=$1 CTTA_GN_APPLY_FUSED=$2 python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/bench_f$1_a$2_$RANDOM.json 2>> $O/bench.err
done
for f in $O/bench_f*.json; do python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d.get("eager_clips_per_s"), d.get("stage_ms"), d["roofline"]["frac"])
PY
done > $O/summary.txt 2>&1
