#!/bin/bash
set -u
R=$PWD
O=$R/gpurun_out/p20
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "chained_resblock or fused_resblock" > $O/test.txt 2>&1
tail -3 $O/test.txt
python3 tools/reschain_bench.py 2>&1 | tail -4 | tee $O/chain.txt
for a in "32 3 1" "64 7 5" "128 11 5"; do python3 tools/resunit_timeline.py $a 2>&1 | grep -v amdgpu.ids >> $O/timeline.txt; done
cat $O/timeline.txt
timeout 900 python3 -m pytest tests/test_engines_gpu.py -x -q -k "hifigan or vocoder or vae" > $O/test_eng.txt 2>&1
tail -3 $O/test_eng.txt
cd /tmp; export TMPDIR=/tmp
for c in 0 0; do
  python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/gen_c$c.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('$O/gen_c$c.json').read().strip().splitlines()[-1]);print('gen', d['value'], d['stage_ms'], d['roofline']['frac'])" >> $O/gen.txt
done
cat $O/gen.txt
