#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p32
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_distill -o p -- python3 $R/bench.py --mode distill --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_distill.log 2>&1
db=$(find $O/prof_distill -name '*.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db $O/rocprof_stats_distill.md > /dev/null
rm -rf $O/prof_distill
grep -v "conv_gemm_kernel\|attention\|attn_bwd\|wgrad_implicit" $O/rocprof_stats_distill.md | head -40 | cut -c1-150
