#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/r5c14; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in 512 256 512 256; do
  CTTA_BIG_TILE_MIN_K=$v python3 $R/bench.py --mode gen --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('BIG_TILE_MIN_K=$v', d['value'], d['stage_ms_graph'], d['roofline']['frac'])"
done | tee $O/gen.txt
for v in 512 256 512 256; do
  CTTA_BIG_TILE_MIN_K=$v python3 $R/bench.py --mode teacher --teacher-steps 100 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('BIG_TILE_MIN_K=$v teacher q/s', d['unet_queries_per_s'])"
done | tee $O/teacher.txt
