#!/bin/bash
# round 5: `python3 bench.py --gpus 2` by itself (no torchrun around it), two ranks sharing the one GPU, gloo transport
set -u
R=$PWD
O=$R/gpurun_out/r5c8; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
( time CTTA_BENCH_BACKEND=gloo timeout 2400 python3 $R/bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench2.json 2> $O/bench2.err ) 2> $O/time.txt
echo "rc=$?" >> $O/time.txt
tail -4 $O/time.txt
tail -5 $O/bench2.err
tail -1 $O/bench2.json | cut -c1-1500
