#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p21
rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -k "groupnorm or gn" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for c in 0 1; do
  python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/gen_$c.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('$O/gen_$c.json').read().strip().splitlines()[-1]);print('gen', d['value'], d['stage_ms'], d['roofline']['frac'])" >> $O/gen.txt
done
cat $O/gen.txt
python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/distill.json
python3 -c "
import json
d=json.loads(open('$O/distill.json').read());print('distill', d['ms_per_step'], d.get('eager_ms_per_step'), d['roofline'])"
python3 $R/bench.py --mode teacher --no-cpu-baseline 2>/dev/null | tail -1 > $O/teacher.json
python3 -c "
import json
d=json.loads(open('$O/teacher.json').read());print('teacher', d['value'], d['ms_per_step'])"
