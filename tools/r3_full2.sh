#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/full2
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu > $O/tests.txt 2>&1
echo "pytest rc=$?" >> $O/tests.txt
cd /tmp; export TMPDIR=/tmp
for t in 0 1 0 1; do
  CTTA_TAIL_SPLIT=$t python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/bench_tail$t_$RANDOM.json 2>> $O/bench.err
done
for f in $O/bench_tail*.json; do python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d.get("eager_clips_per_s"), d.get("stage_ms"), d["roofline"]["frac"])
PY
done > $O/summary.txt 2>&1
