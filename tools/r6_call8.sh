#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/r6_call8; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_clap_gpu.py -q -m gpu -s -k "clap_loss_end_to_end" 2>&1 | grep -i "rel_l2\|passed\|failed" | tee $O/clap.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 python3 $R/bench.py --mode perceptual --no-cpu-baseline --steps 5 --warmup 2 2> $O/perc.err | tail -1 > $O/perc.json
python3 - $O/perc.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ms_per_step", "eager_ms_per_step", "pipelined_ms_per_step", "samples_per_s", "fused_micro_batch")})
PY
tail -3 $O/perc.err
for i in 1 2; do
timeout 900 python3 $R/bench.py --mode distill --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/d.json
python3 - $O/d.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ms_per_step", "eager_ms_per_step", "segmented_pipelined_ms_per_step", "graph_ms_per_step", "pipelined_ms_per_step")})
PY
done
timeout 900 python3 $R/bench.py --mode gen --no-cpu-baseline 2>/dev/null | tail -1 > $O/g.json
python3 - $O/g.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stage_ms_graph")}, d["roofline"]["frac"])
PY
