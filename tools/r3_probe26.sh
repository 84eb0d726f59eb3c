#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
for c in 0 4096 1024 0 4096; do
  CTTA_THIN_RING=$c python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/d.json
  python3 -c "
import json
d=json.loads(open('/tmp/d.json').read());print('ring$c distill', d['ms_per_step'], d.get('eager_ms_per_step'), d['roofline']['kernel_ms_per_step'])"
done
for c in 0 4096 1024; do
  CTTA_THIN_RING=$c python3 $R/bench.py --mode teacher --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/t.json
  python3 -c "
import json
d=json.loads(open('/tmp/t.json').read());print('ring$c teacher', d['teacher']['unet_queries_per_s'] if 'teacher' in d else d)" | cut -c1-300
done
