#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p23
rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_train_gpu.py -x -q > $O/train_test.txt 2>&1; tail -3 $O/train_test.txt | cut -c1-200
cd /tmp; export TMPDIR=/tmp
for c in 1 0; do
  CTTA_WGRAD_IMPLICIT=$c python3 $R/bench.py --mode distill --no-cpu-baseline --no-latency --steps 4 --warmup 2 --profile-csv $O/prof_$c.csv > /dev/null 2>&1
  python3 $R/tools/launch_table.py $O/prof_$c.csv 400 4 > $O/table_$c.txt 2>&1
done
grep -E "^0 +(13[01]|11[0-9]|12[0-9]) " $O/table_1.txt | head -30
echo ---
grep -E "^0 +(13[01]|11[0-9]|12[0-9]) " $O/table_0.txt | head -30
