#!/bin/bash
set -u
R=$PWD
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "groupnorm or gn or concat" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
for c in 1 2 3; do
  python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 8 2>&1 | tail -1
done
python3 $R/tools/prof_unet.py --batch 32 --guided 1 --iters 8 2>&1 | tail -1
