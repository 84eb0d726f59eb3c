#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/r5c19; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_bwd_ops_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/ops.txt
timeout 2400 python3 -m pytest tests/test_train_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/train.txt
cd /tmp; export TMPDIR=/tmp
export CTTA_BENCH_DISTILL_FORMS=pipe
run() { python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['eager_ms_per_step'])"; }
CTTA_WGRAD_DIRECT=0 run "direct=0"
CTTA_WGRAD_DIRECT=1 run "direct=1"
CTTA_WGRAD_DIRECT=0 run "direct=0"
CTTA_WGRAD_DIRECT=1 run "direct=1"
