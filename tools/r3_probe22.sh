#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p22
rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
for c in 1 0 1 0; do
  CTTA_WGRAD_IMPLICIT=$c python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/distill_$c.json
  python3 -c "
import json
d=json.loads(open('$O/distill_$c.json').read());print('implicit$c distill', d['ms_per_step'], d.get('eager_ms_per_step'), d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
done
