#!/bin/bash
# the two PMC passes of the distillation leg alone (each bounded: they hang intermittently under --pmc), summary kept even if one fails
set -u
R=$PWD
O=$R/gpurun_out/pmcd
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_d -o p -- python3 $R/bench.py --mode distill --steps 1 --warmup 1 --no-cpu-baseline --no-latency > $O/pmc_fetch_d.log 2>&1
for try in 1 2; do
  rm -rf $O/pmc_write_d
  timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_d -o p -- python3 $R/bench.py --mode distill --steps 1 --warmup 1 --no-cpu-baseline --no-latency > $O/pmc_write_d.log 2>&1
  ls $O/pmc_write_d/*counter_collection.csv > /dev/null 2>&1 && break
done
python3 $R/tools/pmc_traffic.py $O/pmc_fetch_d $O/pmc_write_d distill > $O/pmc_traffic_distill.json 2> $O/pmc_traffic_distill.err
rm -rf $O/pmc_fetch_d $O/pmc_write_d
cat $O/pmc_traffic_distill.json | head -60; cat $O/pmc_traffic_distill.err | tail -3
