#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/full8
rm -rf $O; mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) > $O/tests.txt 2>&1
echo "pytest rc=$?" >> $O/tests.txt
( time python3 -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
cd /tmp; export TMPDIR=/tmp
( time timeout 1200 python3 $R/bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$?" >> $O/bench.err
