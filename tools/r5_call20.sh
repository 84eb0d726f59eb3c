#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/r5c20; rm -rf $O; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_train_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/train.txt
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
CTTA_WGRAD_DIRECT=$v CTTA_BENCH_DISTILL_FORMS=pipe rocprofv3 --kernel-trace -d $O/prof$v -o p -- python3 $R/bench.py --mode distill --steps 6 --warmup 2 --no-cpu-baseline > $O/prof$v.log 2>&1
db=$(find $O/prof$v -name '*.db' | head -1)
python3 $R/tools/rocpd_gaps.py $db $O/gaps$v.txt adamw:11:16 > /dev/null 2>&1
echo "== direct=$v"; head -8 $O/gaps$v.txt | tail -6; grep -E "wgrad_scatter_rows|wgrad_tn_kernel|wgrad_implicit|col_scatter|splitk_finish" $O/gaps$v.txt | grep " x " | head -8
rm -rf $O/prof$v
done
