#!/bin/bash
set -u
R=$PWD
cd /tmp; export TMPDIR=/tmp
for c in 1 0 1 0; do
  CTTA_GN_FINALIZE_WIDE=$c python3 $R/bench.py --mode gen --no-cpu-baseline --steps 10 > /tmp/g.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('/tmp/g.json').read().strip().splitlines()[-1]);print('wide$c gen', d['value'], d['ms_per_step'], d['stage_ms'])"
done
