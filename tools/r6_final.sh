#!/bin/bash
# Round-6 closing call: the whole GPU suite, smoke, then tools/refresh_profiles.sh (default bench line, kernel tables, launch tables,
# PMC passes, gap accounting) -- ONE call, one box, one code state.
set -u
R=$PWD; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
( time timeout 2400 python3 -m pytest tests -q -m gpu ) > $O/tests.txt 2>&1; grep -E "^FAILED|^ERROR|passed|failed" $O/tests.txt | tail -10
( time python3 -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; tail -2 $O/refresh.log
cp -r $R/gpurun_out/refresh/* $O/ 2>/dev/null
tail -c 1500 $O/bench.json
