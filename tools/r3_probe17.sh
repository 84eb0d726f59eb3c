#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p17
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for r in 3 2; do for b in 16 32 9 18; do
  echo -n "rules$r " >> $O/unet.txt; CTTA_TILE_RULES=$r python3 $R/tools/prof_unet.py --batch $b --guided 0 --iters 8 2>&1 | tail -1 >> $O/unet.txt
done; done; done
for r in 3 2 3 2; do
  CTTA_TILE_RULES=$r python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/gen_r$r.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('$O/gen_r$r.json').read().strip().splitlines()[-1]);print('rules$r gen', d['value'], d['stage_ms'], d['roofline']['frac'])" >> $O/gen.txt
done
