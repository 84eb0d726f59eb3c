#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p27
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_bwd_ops_gpu.py -x -q 2>&1 | tail -2
timeout 1500 python3 -m pytest tests/test_train_gpu.py -x -q > $O/train.txt 2>&1; grep -E "passed|failed" $O/train.txt | tail -1
cd /tmp; export TMPDIR=/tmp
for c in 0 0; do
  python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/distill.json
  python3 -c "
import json
d=json.loads(open('$O/distill.json').read());print('distill', d['ms_per_step'], d.get('eager_ms_per_step'), d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
done
rocprofv3 --kernel-trace --stats -d $O/prof_distill -o p -- python3 $R/bench.py --mode distill --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_distill.log 2>&1
db=$(find $O/prof_distill -name '*.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db $O/rocprof_stats_distill.md > /dev/null
python3 $R/bench.py --mode distill --steps 3 --warmup 1 --no-cpu-baseline --profile-csv $O/launch_distill.csv > /dev/null 2>&1
python3 $R/tools/launch_table.py $O/launch_distill.csv.distill 60 1 > $O/launch_table_distill.txt 2>&1
rm -rf $O/prof_distill $O/launch_distill.csv*
head -30 $O/rocprof_stats_distill.md | cut -c1-130
