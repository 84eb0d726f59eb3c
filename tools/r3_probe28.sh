#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p28
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "tile or conv" 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_engines_gpu.py tests/test_models_gpu.py -x -q > $O/eng.txt 2>&1; grep -E "passed|failed" $O/eng.txt | tail -1
cd /tmp; export TMPDIR=/tmp
for c in 1 0 1 0; do
  CTTA_THIN_RING=$c python3 $R/bench.py --mode teacher --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/t.json
  python3 -c "
import json
d=json.loads(open('/tmp/t.json').read());print('ring$c teacher', d['unet_queries_per_s'])"
done
for c in 1 0 1 0; do
  CTTA_THIN_RING=$c python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/d.json
  python3 -c "
import json
d=json.loads(open('/tmp/d.json').read());print('ring$c distill', d['ms_per_step'], d.get('eager_ms_per_step'))"
done
for c in 1 0 1 0; do
  CTTA_THIN_RING=$c python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > /tmp/g.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('/tmp/g.json').read().strip().splitlines()[-1]);print('ring$c gen', d['value'], d['stage_ms'])"
done
