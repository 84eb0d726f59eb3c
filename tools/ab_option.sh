#!/bin/bash
# Same-box A/B of one library option through bench.py: alternates CTTA_OPT_<NAME>=<off> and the default, ROUNDS times each.
# usage: tools/ab_option.sh <NAME> <off value> <rounds> [bench args...]      e.g. tools/ab_option.sh FFN_FUSE 0 2 --mode distill
NAME=$1; OFF=$2; ROUNDS=$3; shift 3
for r in $(seq 1 $ROUNDS); do
  for v in "$OFF" default; do
    if [ "$v" = default ]; then out=$(python bench.py "$@" 2>/dev/null | tail -1); else out=$(env CTTA_OPT_$NAME=$v python bench.py "$@" 2>/dev/null | tail -1); fi
    echo "$NAME=$v: $(echo "$out" | python -c '
import json, sys
d = json.loads(sys.stdin.read())
def pick(o, pre=""):
    r = []
    for k in ("value", "ms_per_step", "stage_ms_graph", "eager_ms_per_step", "pipelined_ms_per_step", "graph_ms_per_step"):
        if isinstance(o, dict) and o.get(k) is not None:
            r.append("%s%s=%s" % (pre, k, o[k]))
    return r
out = pick(d)
for leg in ("distill", "teacher", "perceptual_distill"):
    out += pick(d.get(leg), leg + ".")
print("  ".join(out))')"
  done
done
