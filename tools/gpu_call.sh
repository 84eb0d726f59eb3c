#!/bin/bash
# One parameterised GPU-box script (replaces round 3's fifty one-off r3_probe*.sh / r3_full*.sh):
#   gpurun --timeout 2400 -- 'bash tools/gpu_call.sh full'                 # whole GPU suite + smoke + default bench
#   gpurun --timeout 900  -- 'bash tools/gpu_call.sh ab CTTA_OPT_STREAMK distill 1 0 1 0'   # A/B a library option (CTTA_OPT_<NAME>, applied by _native at load) on a bench mode
#   gpurun --timeout 600  -- 'bash tools/gpu_call.sh pmc_distill'          # FETCH / WRITE PMC passes of the distillation leg
#   gpurun --timeout 600  -- 'bash tools/gpu_call.sh tests "-k segmented" tests/test_train_gpu.py'
# Profiles for profiles/ come from tools/refresh_profiles.sh.
set -u
R=$PWD
what=${1:-full}; shift || true
O=$R/gpurun_out/$what
rm -rf $O; mkdir -p $O
case $what in
  full)
    ( time timeout 3000 python3 -m pytest tests -x -q -m gpu ) > $O/tests.txt 2>&1
    echo "pytest rc=$?" >> $O/tests.txt
    ( time python3 -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
    echo "smoke rc=$?" >> $O/smoke.txt
    cd /tmp; export TMPDIR=/tmp
    ( time timeout 1200 python3 $R/bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
    echo "bench rc=$?" >> $O/bench.err
    tail -3 $O/tests.txt; tail -2 $O/smoke.txt; tail -c 600 $O/bench.json
    ;;
  tests)
    sel=${1:-}; shift || true
    timeout 2400 python3 -m pytest ${@:-tests} -x -q -m gpu $sel 2>&1 | tail -25 | tee $O/tests.txt
    ;;
  ab)   # ab <ENV_NAME> <bench mode> <value> [<value> ...]: one bench run per value, in the given order (repeat values for noise)
    var=$1; mode=$2; shift 2
    cd /tmp; export TMPDIR=/tmp
    for v in "$@"; do
      env_line="$var=$v"
      export "$var=$v"
      python3 $R/bench.py --mode $mode --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/ab.json
      python3 - "$env_line" $O/ab.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
keys = ("value", "ms_per_step", "eager_ms_per_step", "segmented_ms_per_step", "unet_queries_per_s", "stage_ms", "stage_ms_graph")
print(sys.argv[1], {k: d[k] for k in keys if k in d})
PY
    done | tee $O/ab.txt
    ;;
  pmc_distill)
    cd /tmp; export TMPDIR=/tmp
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_d -o p -- python3 $R/bench.py --mode distill --steps 1 --warmup 1 --no-cpu-baseline --no-latency > $O/pmc_fetch_d.log 2>&1
    for try in 1 2; do
      rm -rf $O/pmc_write_d
      timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_d -o p -- python3 $R/bench.py --mode distill --steps 1 --warmup 1 --no-cpu-baseline --no-latency > $O/pmc_write_d.log 2>&1
      ls $O/pmc_write_d/*counter_collection.csv > /dev/null 2>&1 && break
    done
    python3 $R/tools/pmc_traffic.py $O/pmc_fetch_d $O/pmc_write_d distill > $O/pmc_traffic_distill.json 2> $O/pmc_traffic_distill.err
    rm -rf $O/pmc_fetch_d $O/pmc_write_d
    head -60 $O/pmc_traffic_distill.json; tail -3 $O/pmc_traffic_distill.err
    ;;
  sweep_ab)   # sweep_ab <ENV_NAME> <SWEEP_FILTER> <value> [<value> ...]: tools/sweep_conv.py (auto variant only, cold weights) per value
    var=$1; flt=$2; shift 2
    for v in "$@"; do
      echo "== $var=$v"
      env "$var=$v" SWEEP_FILTER="$flt" SWEEP_COLD=${SWEEP_COLD:-1} SWEEP_BRIEF=1 SWEEP_VARIANTS=${SWEEP_VARIANTS:-22} python3 $R/tools/sweep_conv.py
    done 2>&1 | grep -v "^variants" | tee $O/sweep_ab.txt
    ;;
  *) echo "unknown mode $what"; exit 2 ;;
esac
