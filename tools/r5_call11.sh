#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/r5c11; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/ops.txt
cd /tmp; export TMPDIR=/tmp
mkdir -p $O/slab1
CTTA_XCD_SLAB=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/slab1 -o p -- python3 $R/tools/pmc_thin_shapes.py run $O/slab1/manifest.json > $O/slab1/log.txt 2>&1
python3 $R/tools/pmc_thin_shapes.py parse $O/slab1 2>&1 | tee $O/pmc_by_shape_thin_b.txt
rm -rf $O/slab1/p* $O/slab1/*/ 2>/dev/null
for v in 1 2 1 2; do
  CTTA_THIN_RING=$v python3 $R/bench.py --mode teacher --teacher-steps 100 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('THIN_RING=$v teacher q/s', d['unet_queries_per_s'])"
done | tee $O/teacher.txt
export CTTA_BENCH_DISTILL_FORMS=pipe
for v in 1 2 1 2; do
  CTTA_THIN_RING=$v python3 $R/bench.py --mode distill --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('THIN_RING=$v distill', d['ms_per_step'], d['eager_ms_per_step'])"
done | tee $O/distill.txt
