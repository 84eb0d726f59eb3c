#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p40
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/prof_unet.py --batch 32 --guided 1 --iters 8 --profile-csv $O/unet.csv > $O/unet_log.txt 2>&1
python3 $R/tools/launch_table.py $O/unet.csv 60 1 > $O/launch_table_unet.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 $R/tools/prof_unet.py --batch 32 --guided 1 --iters 8 > $O/log.txt 2>&1
db=$(find $O/prof -name '*.db' | head -1)
python3 $R/tools/rocpd_timeline.py $db $O/unet_timeline.txt 1500 > /dev/null 2>&1
rm -rf $O/prof $O/unet.csv
head -3 $O/launch_table_unet.txt; head -12 $O/unet_timeline.txt; tail -1 $O/unet_log.txt
