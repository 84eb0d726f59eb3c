#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p6
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -k "groupnorm or attention" > $O/tests.txt 2>&1
CTTA_ATTN_V2=0 python3 tools/attn_bench.py > $O/attn_v0.txt 2>&1
CTTA_ATTN_V2=1 python3 tools/attn_bench.py > $O/attn_v1.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for t in 0 1 0 1; do
  CTTA_ATTN_V2=0 CTTA_TAIL_SPLIT=$t python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/bench_tail${t}_$RANDOM.json 2>> $O/bench.err
done
CTTA_ATTN_V2=1 python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/bench_attnv2_$RANDOM.json 2>> $O/bench.err
for f in $O/bench_*.json; do python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d.get("eager_clips_per_s"), d.get("stage_ms"), d["roofline"]["frac"])
PY
done > $O/summary.txt 2>&1
