#!/bin/bash
# round 5, call 5: the touched tests + A/B of the thin-tile rules and the slab mapping on the distillation leg
set -u
R=$PWD
O=$R/gpurun_out/r5c5; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_bwd_ops_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/ops.txt
timeout 1500 python3 -m pytest tests/test_train_gpu.py -x -q -m gpu -k "pipelined or full_batch or segmented" 2>&1 | tail -8 | tee $O/train.txt
cd /tmp; export TMPDIR=/tmp
export CTTA_BENCH_FUSED_ACCUM=0
for cfg in "0 0" "1 0" "1 1" "0 1" "1 1" "0 0"; do
  set -- $cfg
  CTTA_TILE_RULES_R5=$1 CTTA_XCD_SLAB=$2 python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/ab.json
  python3 - "R5=$1 SLAB=$2" $O/ab.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
keys = ("ms_per_step", "eager_ms_per_step", "graph_ms_per_step", "segmented_pipelined_ms_per_step", "fixed_draw_loss_before_after")
print(sys.argv[1], {k: d.get(k) for k in keys}, "frac", d.get("roofline", {}).get("frac"), "kernel_ms", d.get("roofline", {}).get("kernel_ms_per_step"))
PY
done | tee $O/ab.txt
for v in "0 0" "1 1"; do
  set -- $v
  CTTA_TILE_RULES_R5=$1 CTTA_XCD_SLAB=$2 python3 $R/bench.py --mode teacher --teacher-steps 100 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['unet_queries_per_s'])"
done | tee $O/teacher.txt
