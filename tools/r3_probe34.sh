#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
for c in 1280 2048 2304 1280 2048; do
  CTTA_SPLITK_M64=$c python3 $R/tools/prof_unet.py --batch 32 --guided 1 --iters 8 2>&1 | tail -1 | sed "s/^/m64=$c unet32 /"
done
for c in 1280 2304 1280 2304; do
  CTTA_SPLITK_M64=$c python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/d.json
  python3 -c "
import json
d=json.loads(open('/tmp/d.json').read());print('m64=$c distill', d['ms_per_step'], d.get('eager_ms_per_step'))"
done
