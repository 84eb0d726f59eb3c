#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
export CTTA_BENCH_DISTILL_FORMS=pipe
run() { python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['eager_ms_per_step'])"; }
run "default (wgrad 256/32/256, splitk 512)"
CTTA_SPLITK_TARGET=384 run "splitk_target 384"
CTTA_SPLITK_TARGET=256 run "splitk_target 256"
CTTA_SPLITK_TARGET=256 CTTA_TILE_RULES_R5=0 run "splitk_target 256, r4 thin rules"
CTTA_SPLITK_TILES=96 run "splitk gate 96 tiles"
run "default again"
