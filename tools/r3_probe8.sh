#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p8
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python3 $R/bench.py --mode distill --no-cpu-baseline --no-latency --steps 10 --warmup 3 > $O/d_$tag.json 2>> $O/err.txt; python3 - $O/d_$tag.json $tag <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["roofline"]["frac"], d.get("grad_accum_5",{}).get("ms_per_optimizer_step"))
PY
}
run all_on A=1 >> $O/summary.txt
run gnfuse0 CTTA_GN_FUSE=0 >> $O/summary.txt
run attn0 CTTA_ATTN_V2=0 >> $O/summary.txt
run all_on2 A=1 >> $O/summary.txt
run both0 CTTA_GN_FUSE=0 CTTA_ATTN_V2=0 >> $O/summary.txt
python3 $R/tools/overlap_model.py $O/overlap_model.json > $O/overlap.log 2>&1
