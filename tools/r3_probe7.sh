#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p7
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -k "groupnorm or attention" > $O/tests.txt 2>&1
CTTA_ATTN_V2=0 python3 tools/attn_bench.py > $O/attn_v0.txt 2>&1
CTTA_ATTN_V2=1 python3 tools/attn_bench.py > $O/attn_v1.txt 2>&1
CTTA_ATTN_V2=0 python3 tools/attn_bench.py >> $O/attn_v0.txt 2>&1
CTTA_ATTN_V2=1 python3 tools/attn_bench.py >> $O/attn_v1.txt 2>&1
