#!/bin/bash
# round 5: (a) FETCH_SIZE per launch of the thin shapes, plain grid vs weight-slab mapping; (b) split-K gates on the mid-K linears;
# (c) the single-read softmax
set -u
R=$PWD
O=$R/gpurun_out/r5c9; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for slab in 0 1; do
  mkdir -p $O/slab$slab
  CTTA_XCD_SLAB=$slab timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/slab$slab -o p -- python3 $R/tools/pmc_thin_shapes.py run $O/slab$slab/manifest.json > $O/slab$slab/log.txt 2>&1
done
python3 $R/tools/pmc_thin_shapes.py parse $O/slab0 $O/slab1 2>&1 | tee $O/pmc_by_shape_thin.txt
rm -rf $O/slab0/p* $O/slab1/p* $O/slab0/*/ $O/slab1/*/ 2>/dev/null
cd $R
for cfg in "32 8" "16 8" "16 4" "8 4"; do
  set -- $cfg
  echo "== CTTA_SPLITK_MIN_NK=$1 CTTA_SPLITK_MIN_STEPS=$2"
  CTTA_SPLITK_MIN_NK=$1 CTTA_SPLITK_MIN_STEPS=$2 SWEEP_FILTER="thin lin" SWEEP_COLD=1 SWEEP_BRIEF=1 SWEEP_VARIANTS=22 python3 tools/sweep_conv.py
  CTTA_SPLITK_MIN_NK=$1 CTTA_SPLITK_MIN_STEPS=$2 SWEEP_FILTER="d9 lin" SWEEP_COLD=1 SWEEP_BRIEF=1 SWEEP_VARIANTS=22 python3 tools/sweep_conv.py
done 2>&1 | grep -v "amdgpu.ids\|^variants" | tee $O/sweep_splitk_gates.txt
timeout 600 python3 -m pytest tests/test_ops_gpu.py tests/test_engines_gpu.py -x -q -m gpu -k "softmax or vae" 2>&1 | tail -4 | tee $O/softmax_tests.txt
cd /tmp
for v in 0 1 0 1; do
  CTTA_SOFTMAX_REG=$v python3 $R/bench.py --mode gen --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('SOFTMAX_REG=$v', d['value'], d['stage_ms_graph'], d['stage_ms'])"
done | tee $O/softmax_ab.txt
