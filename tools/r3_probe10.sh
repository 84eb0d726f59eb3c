#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p10
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_engines_gpu.py tests/test_ops_gpu.py tests/test_bwd_ops_gpu.py tests/test_models_gpu.py -q -m gpu -x > $O/tests.txt 2>&1
echo "rc=$?" >> $O/tests.txt
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/prof_unet.py --batch 32 --iters 5 > $O/unet.log 2>&1
CTTA_GN_SMALL=0 CTTA_CONV_OUT_DIRECT=1 python3 $R/tools/prof_unet.py --batch 32 --iters 5 > $O/unet_old.log 2>&1
python3 $R/tools/prof_unet.py --batch 16 --guided 0 --iters 5 > $O/unet16.log 2>&1
