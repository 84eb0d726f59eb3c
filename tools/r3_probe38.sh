#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p38
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 8 > $O/log.txt 2>&1
db=$(find $O/prof -name '*.db' | head -1)
python3 $R/tools/rocpd_timeline.py $db $O/timeline.txt 1500 > /dev/null 2>&1
rm -rf $O/prof
head -40 $O/timeline.txt
