#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p31
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_bwd_ops_gpu.py -x -q 2>&1 | tail -2
timeout 1500 python3 -m pytest tests/test_train_gpu.py tests/test_clap_gpu.py -x -q > $O/train.txt 2>&1; grep -E "passed|failed" $O/train.txt | tail -1
cd /tmp; export TMPDIR=/tmp
for c in 1 0 1 0; do
  CTTA_GN_BWD_SMALL=$c python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/d.json
  python3 -c "
import json
d=json.loads(open('/tmp/d.json').read());print('new$c distill', d['ms_per_step'], d.get('eager_ms_per_step'))"
done
