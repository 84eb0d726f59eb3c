#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
export CTTA_BENCH_DISTILL_FORMS=pipe
run() { python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['eager_ms_per_step'])"; }
for cfg in "128 16 256" "64 16 128" "128 32 128" "64 8 64" "256 16 256" "128 16 256" "32 8 256" "512 64 512"; do
  set -- $cfg
  CTTA_WGRAD_TN_TARGET=$1 CTTA_WGRAD_TN_MAX_SPLITS=$2 CTTA_WGRAD_CONV_TARGET=$3 run "tn_target=$1 tn_cap=$2 conv_target=$3"
done
