#!/bin/bash
# Regenerates everything under profiles/ that a round's numbers come from.  Run on the GPU box from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# then copy gpurun_out/refresh/* into profiles/ (see profiles/README.md).
set -u
R=$PWD
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
# Kernel tables that belong to ONE launch form each, with the step count in the header (round 5):
#   gen:     CTTA_BENCH_MINIMAL=1 CTTA_BENCH_GRAPH=0 -> 2 x (1 warm-up + 5 timed) + 1 = 13 eager batch-32 steps, nothing else
#   distill: CTTA_BENCH_DISTILL_FORMS=eager          -> 1 + 5 eager optimizer steps at batch 9 (+ 2 loss-only forwards of the fixed draw)
CTTA_BENCH_MINIMAL=1 CTTA_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/prof_gen -o p -- python3 $R/bench.py --mode gen --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_gen.log 2>&1
# (round 6: the distillation table is taken in the SINGLE-STREAM form -- no student side stream, no weight-gradient side stream --
#  so that the kernel times of the table add up to a duration: CTTA_TWO_STREAM=0 is AudioLCM's switch, CTTA_OPT_WGRAD_STREAM=0 the
#  library option "wgrad_stream" as _native applies it at load)
CTTA_TWO_STREAM=0 CTTA_OPT_WGRAD_STREAM=0 CTTA_BENCH_DISTILL_FORMS=eager rocprofv3 --kernel-trace --stats -d $O/prof_distill -o p -- python3 $R/bench.py --mode distill --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_distill.log 2>&1
db=$(find $O/prof_gen -name '*.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db $O/rocprof_stats_gen.md "rocprofv3 --kernel-trace over: CTTA_BENCH_MINIMAL=1 CTTA_BENCH_GRAPH=0 bench.py --mode gen --steps 5 --warmup 1 = 13 eager batch-32 generation steps (2 x (1 + 5) timed-loop steps + 1), model set-up kernels (weight init / pack) included" > /dev/null
db=$(find $O/prof_distill -name '*.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_stats.py $db $O/rocprof_stats_distill.md "rocprofv3 --kernel-trace over: CTTA_BENCH_DISTILL_FORMS=eager bench.py --mode distill --steps 5 --warmup 1 = 6 eager optimizer steps at batch 9, every launch on ONE stream (CTTA_TWO_STREAM=0, option wgrad_stream=0: kernel times add up) + 2 loss-only forward passes of the fixed draw + the single-stream profiled step of the roofline object = 7 step-equivalents of MFMA work, model set-up kernels included" > /dev/null
python3 $R/bench.py --mode gen --steps 3 --warmup 1 --no-cpu-baseline --profile-csv $O/launch_gen.csv > /dev/null 2>&1
python3 $R/tools/launch_table.py $O/launch_gen.csv 60 2 > $O/launch_table_gen.txt 2>&1
CTTA_BENCH_DISTILL_FORMS=none python3 $R/bench.py --mode distill --steps 3 --warmup 1 --no-cpu-baseline --profile-csv $O/launch_distill.csv > /dev/null 2>&1
python3 $R/tools/launch_table.py $O/launch_distill.csv.distill 60 1 > $O/launch_table_distill.txt 2>&1
# REFRESH_LIGHT=1: bench line, kernel stats and launch tables only (no counter passes, no gap accounting): ~5 minutes
if [ "${REFRESH_LIGHT:-0}" = "1" ]; then
  rm -rf $O/prof_gen $O/prof_distill $O/launch_gen.csv $O/launch_distill.csv $O/launch_distill.csv.distill
  du -sh $O
  exit 0
fi
# one counter pass; repeated once when rocprofv3 died before it wrote a counter table (these passes have failed intermittently
# at start-up since round 1: the log then ends after "HSA version ... initialized")
pmc_pass() {   # <counter> <out dir name> <timeout> <bench args...>
  local ctr=$1 dir=$2 to=$3; shift 3
  for try in 1 2; do
    rm -rf $O/$dir
    timeout $to rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$dir -o p -- python3 $R/bench.py "$@" > $O/$dir.log 2>&1
    [ -n "$(find $O/$dir -name '*counter_collection.csv' -size +1k 2>/dev/null | head -1)" ] && return 0
    echo "pmc pass $dir: no counter table (try $try)" >> $O/pmc_retries.txt
  done
  return 1
}
pmc_pass FETCH_SIZE pmc_fetch 420 --mode gen --steps 2 --warmup 1 --no-cpu-baseline --no-latency
pmc_pass WRITE_SIZE pmc_write 420 --mode gen --steps 2 --warmup 1 --no-cpu-baseline --no-latency
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2> $O/pmc_traffic.err
# Distillation leg: the PMC passes of that mode hung intermittently in round 1; one step, no warm-up, no latency legs
# keeps each pass short, and a 300 s timeout bounds the loss when one hangs (tools/pmc_traffic.py ... distill <steps>;
# steps = 1 timed + 1 warm-up minimum + 10 accumulation micro-steps + 1 profiled = see the script's own count)
pmc_pass FETCH_SIZE pmc_fetch_d 300 --mode distill --steps 1 --warmup 1 --no-cpu-baseline --no-latency
pmc_pass WRITE_SIZE pmc_write_d 300 --mode distill --steps 1 --warmup 1 --no-cpu-baseline --no-latency
python3 $R/tools/pmc_traffic.py $O/pmc_fetch_d $O/pmc_write_d distill 18 > $O/pmc_traffic_distill.json 2> $O/pmc_traffic_distill.err
pmc_pass SQ_VALU_MFMA_BUSY_CYCLES pmc_mfma 300 --mode gen --steps 2 --warmup 1 --no-cpu-baseline --no-latency
python3 $R/tools/pmc_mfma_util.py $O/pmc_mfma > $O/pmc_mfma_util.json 2> $O/pmc_mfma_util.err
# where a hipGraph-replayed distillation step spends its time: idle / one kernel / two kernels, dispatch counts, idle gaps by kernel
# pair (three monolithic-graph steps between AdamW launches: eager 1+3, segmented 1+3, monolithic 1+3 -> AdamW launches 8..11)
# (round 5: the PIPELINED monolithic form, the one the headline distillation number runs: eager 2 + 6 optimizer steps, then the
# pipelined capture's 1 + 2 + 6 -> AdamW launches 11..16 bracket five clean replayed steps)
CTTA_BENCH_DISTILL_FORMS=pipe rocprofv3 --kernel-trace -d $O/prof_gaps -o p -- python3 $R/bench.py --mode distill --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_gaps.log 2>&1
db=$(find $O/prof_gaps -name '*.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_gaps.py $db $O/gaps_distill_pipelined.txt adamw:11:16 > /dev/null 2> $O/gaps_distill.err
rm -rf $O/prof_gen $O/prof_distill $O/prof_gaps $O/launch_gen.csv $O/launch_distill.csv $O/launch_distill.csv.distill
# the raw counter CSVs are ~85 MB together and gpurun copies back at most 64 MiB: only the summaries travel
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_fetch_d $O/pmc_write_d $O/pmc_mfma
du -sh $O
