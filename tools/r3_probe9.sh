#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p9
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/prof_unet.py --batch 32 --iters 5 --profile-csv $O/unet_launch.csv > $O/unet.log 2>&1
python3 $R/tools/launch_table.py $O/unet_launch.csv 80 1 > $O/unet_launch_table.txt 2>&1
python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 5 --profile-csv $O/unet16_launch.csv > $O/unet16.log 2>&1
python3 $R/tools/launch_table.py $O/unet16_launch.csv 40 1 > $O/unet16_launch_table.txt 2>&1
rocprofv3 --kernel-trace -d $O/prof_unet -o p -- python3 $R/tools/prof_unet.py --batch 32 --iters 2 > $O/prof_unet.log 2>&1
db=$(find $O/prof_unet -name '*.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_timeline.py $db $O/unet_timeline.txt 1500
rm -rf $O/prof_unet
