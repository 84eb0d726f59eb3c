#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/full6
rm -rf $O; mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) > $O/tests.txt 2>&1
echo "pytest rc=$?" >> $O/tests.txt
( time python3 -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
cd /tmp; export TMPDIR=/tmp
( time timeout 1200 python3 $R/bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$?" >> $O/bench.err
O=$R/gpurun_out/full6
rocprofv3 --kernel-trace --stats -d $O/prof_distill -o p -- python3 $R/bench.py --mode distill --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_distill.log 2>&1
db=$(find $O/prof_distill -name '*.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db $O/rocprof_stats_distill.md > /dev/null
python3 $R/bench.py --mode distill --steps 3 --warmup 1 --no-cpu-baseline --profile-csv $O/launch_distill.csv > /dev/null 2>&1
python3 $R/tools/launch_table.py $O/launch_distill.csv.distill 60 1 > $O/launch_table_distill.txt 2>&1
rm -rf $O/prof_distill $O/launch_distill.csv*
