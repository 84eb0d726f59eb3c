#!/bin/bash
# full GPU check: every -m gpu test, then the default bench line
set -u
R=$PWD
O=$R/gpurun_out/full
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
echo "pytest rc=$?" >> $O/tests.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
