#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p25
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "concat_with or groupnorm" 2>&1 | tail -4
timeout 1500 python3 -m pytest tests/test_engines_gpu.py -x -q > $O/eng.txt 2>&1; grep -E "passed|failed" $O/eng.txt | tail -2
cd /tmp; export TMPDIR=/tmp
for c in 1 0 1 0; do
  CTTA_GN_FUSE=$c python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/gen_$c.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('$O/gen_$c.json').read().strip().splitlines()[-1]);print('gnfuse$c gen', d['value'], d['stage_ms'], d['roofline']['frac'])"
done
python3 $R/bench.py --mode teacher --no-cpu-baseline 2>/dev/null | tail -1 > $O/teacher.json
python3 -c "
import json
d=json.loads(open('$O/teacher.json').read());print('teacher', {k: d[k] for k in d if k in ('value','unit','teacher')})" | cut -c1-600
