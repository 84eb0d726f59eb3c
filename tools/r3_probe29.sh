#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
for c in 1 0 1 0; do
  CTTA_THIN_RING=$c python3 $R/bench.py --mode teacher --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/t.json
  python3 -c "
import json
d=json.loads(open('/tmp/t.json').read());print('ring$c teacher', d['unet_queries_per_s'])"
done
CTTA_THIN_RING=1 python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/d.json
python3 -c "
import json
d=json.loads(open('/tmp/d.json').read());print('ring1 distill', d['ms_per_step'], d.get('eager_ms_per_step'))"
cd $R; timeout 900 python3 -m pytest tests/test_models_gpu.py tests/test_engines_gpu.py -x -q 2>&1 | tail -2
