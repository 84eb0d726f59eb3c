#!/bin/bash
# A/B of conv_gemm tuning knobs on the distillation step: tools/ab_env.sh "CTTA_SPLITK_TARGET=512 CTTA_SPLITK_MAX=8" "..." ...
for cfg in "$@"; do
  ms=$(env $cfg python3 bench.py --mode distill --steps 6 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$cfg -> $ms ms"
done
