#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
for cfg in "CTTA_SPLITK_M64=1280" "CTTA_SPLITK_M64=2048" "CTTA_SPLITK_M64=1280" "CTTA_SPLITK_M64=2048"; do
  env $cfg python3 $R/tools/prof_unet.py --batch 32 --guided 1 --iters 8 2>&1 | tail -1 | sed "s/^/$cfg /"
done
for cfg in "CTTA_SPLITK_M64=1280" "CTTA_SPLITK_M64=2048"; do
  env $cfg python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 8 2>&1 | tail -1 | sed "s/^/$cfg /"
  env $cfg python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/d.json
  python3 -c "
import json
d=json.loads(open('/tmp/d.json').read());print('$cfg distill', d['ms_per_step'])"
done
