#!/bin/bash
set -u
R=$PWD
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "groupnorm or gn or concat" 2>&1 | tail -1
timeout 900 python3 -m pytest tests/test_engines_gpu.py -x -q 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
for c in 1 2 3; do
  python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > /tmp/g.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('/tmp/g.json').read().strip().splitlines()[-1]);print('gen', d['value'], d['stage_ms'], d['roofline']['frac'])"
done
