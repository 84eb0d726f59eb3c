#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/r5c15; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export CTTA_BENCH_DISTILL_FORMS=pipe
run() { python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['eager_ms_per_step'])"; }
run "default(tn 64/512, conv 16/512)"
CTTA_WGRAD_TN_MAX_SPLITS=16 run "tn_cap16"
CTTA_WGRAD_TN_MAX_SPLITS=32 run "tn_cap32"
CTTA_WGRAD_TN_TARGET=256 run "tn_target256"
CTTA_WGRAD_TN_TARGET=128 CTTA_WGRAD_TN_MAX_SPLITS=32 run "tn_target128_cap32"
CTTA_WGRAD_CONV_TARGET=256 run "conv_target256"
CTTA_WGRAD_CONV_TARGET=128 run "conv_target128"
CTTA_WGRAD_TN_TARGET=256 CTTA_WGRAD_CONV_TARGET=256 run "both256"
run "default again"
