#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/r5c12; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_bwd_ops_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/ops.txt
timeout 1500 python3 -m pytest tests/test_train_gpu.py -x -q -m gpu -k "autograd or light_widths or full_batch or adamw" 2>&1 | tail -4 | tee $O/train.txt
cd /tmp; export TMPDIR=/tmp
CTTA_BENCH_DISTILL_FORMS=pipe rocprofv3 --kernel-trace -d $O/prof -o p -- python3 $R/bench.py --mode distill --steps 6 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
tail -1 $O/prof.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined ms/step (profiled)', d['ms_per_step'])"
db=$(find $O/prof -name '*.db' | head -1)
python3 $R/tools/rocpd_gaps.py $db $O/gaps.txt adamw:11:16 > /dev/null 2>&1
grep -n "gn_bwd_small\|gn_bwd_param" $O/gaps.txt | head
rm -rf $O/prof
for i in 1 2; do
CTTA_BENCH_DISTILL_FORMS=pipe python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pipelined ms/step', d['ms_per_step'], d['eager_ms_per_step'])"
done
