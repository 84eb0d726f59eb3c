#!/bin/bash
# round 5: kernel-trace of the PIPELINED monolithic distillation step only (what the headline distill number runs)
set -u
R=$PWD
O=$R/gpurun_out/r5c7; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
CTTA_BENCH_DISTILL_FORMS=pipe rocprofv3 --kernel-trace -d $O/prof -o p -- python3 $R/bench.py --mode distill --steps 6 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
tail -1 $O/prof.log | cut -c1-600
db=$(find $O/prof -name '*.db' | head -1)
python3 $R/tools/rocpd_gaps.py $db $O/gaps_distill_pipelined.txt adamw:11:16 > /dev/null 2> $O/gaps.err
tail -3 $O/gaps.err
rm -rf $O/prof
head -50 $O/gaps_distill_pipelined.txt
