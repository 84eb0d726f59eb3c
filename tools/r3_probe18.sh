#!/bin/bash
# chained ResBlock kernel: parity tests, then A/B of the generation bench with the chain on / off
set -u
R=$PWD
O=$R/gpurun_out/p18
rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "chained_resblock or fused_resblock" > $O/test.txt 2>&1
tail -5 $O/test.txt
timeout 900 python3 -m pytest tests/test_engines_gpu.py -x -q -k "hifigan or vocoder or vae" > $O/test_eng.txt 2>&1
tail -5 $O/test_eng.txt
cd /tmp; export TMPDIR=/tmp
for c in 1 0 1 0; do
  CTTA_RES_CHAIN=$c python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > $O/gen_c$c.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('$O/gen_c$c.json').read().strip().splitlines()[-1]);print('chain$c gen', d['value'], d['stage_ms'], d['roofline']['frac'])" >> $O/gen.txt
done
cat $O/gen.txt
