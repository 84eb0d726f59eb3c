#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p13
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_engines_gpu.py tests/test_models_gpu.py tests/test_train_gpu.py -q -m gpu -x > $O/tests.txt 2>&1
echo "rc=$?" >> $O/tests.txt
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --mode teacher --no-cpu-baseline > $O/teacher.json 2> $O/teacher.err
python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 > $O/distill.json 2> $O/distill.err
