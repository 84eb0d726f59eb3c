#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
export CTTA_BENCH_DISTILL_FORMS=pipe
run() { python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>$R/gpurun_out/r5c17_err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['eager_ms_per_step'], d['teacher_stream_placement_ms'])"; }
run "no masks"
CTTA_TEACHER_CU_MASKS="64,96,128,160,64:spread,128:spread" run "masks"
tail -3 $R/gpurun_out/r5c17_err.txt
