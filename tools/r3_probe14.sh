#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p14
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_ops_gpu.py tests/test_engines_gpu.py -q -m gpu -x -k "layernorm or unet or transformer" > $O/tests.txt 2>&1
echo "rc=$?" >> $O/tests.txt
cd /tmp; export TMPDIR=/tmp
for v in 1 0 1 0; do CTTA_LN_FAST=$v python3 $R/tools/prof_unet.py --batch 32 --iters 8 2>&1 | tail -1 >> $O/unet_ln$v.log; done
CTTA_LN_FAST=1 python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 8 2>&1 | tail -1 >> $O/unet16_ln1.log
CTTA_LN_FAST=0 python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 8 2>&1 | tail -1 >> $O/unet16_ln0.log
