#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/r5c6; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_bwd_ops_gpu.py -x -q -m gpu -k "attention" -s 2>&1 | grep -v "^$" | tail -30 | tee $O/ops.txt
timeout 1500 python3 -m pytest tests/test_train_gpu.py -x -q -m gpu -k "autograd or light_widths or full_batch" 2>&1 | tail -5 | tee $O/train.txt
cd /tmp; export TMPDIR=/tmp
export CTTA_BENCH_FUSED_ACCUM=0
for v in 0 1 0 1; do
  CTTA_ATTN_BWD_INPLACE=$v python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/ab.json
  python3 - "INPLACE=$v" $O/ab.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
keys = ("ms_per_step", "eager_ms_per_step", "graph_ms_per_step", "segmented_pipelined_ms_per_step")
print(sys.argv[1], {k: d.get(k) for k in keys}, "frac", d.get("roofline", {}).get("frac"), "kernel_ms", d.get("roofline", {}).get("kernel_ms_per_step"))
PY
done | tee $O/ab.txt
