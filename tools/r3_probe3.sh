#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p3
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export CTTA_GN_FUSE=1
rocprofv3 --kernel-trace -d $O/prof -o p -- python3 $R/bench.py --mode gen --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $O/prof.log 2>&1
db=$(find $O/prof -name '*.db' | head -1)
python3 $R/tools/rocpd_stats.py $db $O/stats_fuse1.md > /dev/null
python3 $R/tools/rocpd_timeline.py $db $O/timeline_fuse1.txt 4000
rm -rf $O/prof
