#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/sweep
rm -rf $O; mkdir -p $O
cd $R
SWEEP_BRIEF=0 SWEEP_GEGLU=1 SWEEP_FILTER=ff1 SWEEP_VARIANTS=17,18,19,20,21,22,23,24,26,27,28,9,10 python3 tools/sweep_conv.py $O/geglu.json > $O/geglu.log 2>&1
SWEEP_FILTER=u32 SWEEP_VARIANTS=17,18,19,21,22,23,24,26,27,28,29,31,32 python3 tools/sweep_conv.py $O/u32.json > $O/u32.log 2>&1
SWEEP_FILTER=lin SWEEP_VARIANTS=17,18,19,21,22,23,24,26,27,28,29,31,32 python3 tools/sweep_conv.py $O/lin.json > $O/lin.log 2>&1
SWEEP_FILTER=d9 SWEEP_VARIANTS=17,18,19,21,22,23,24,26,27,28,29,31,32 python3 tools/sweep_conv.py $O/d9.json > $O/d9.log 2>&1
SWEEP_EPI=1 SWEEP_FILTER=u32 SWEEP_VARIANTS=17,18,19,21,22,23,24,26,27,28,29,31,32 python3 tools/sweep_conv.py $O/u32_epi.json > $O/u32_epi.log 2>&1
for f in geglu u32 lin d9 u32_epi; do echo "== $f"; python3 tools/sweep_top.py $O/$f.json; done > $O/top.txt 2>&1
