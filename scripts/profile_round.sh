#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats of the default bench run, and per workload the PMC passes
# (FETCH_SIZE / WRITE_SIZE separately, never together with other trace domains).
# usage (on the GPU box, via gpurun): scripts/profile_round.sh r02
R=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
mkdir -p $OUT
# (--no-sweep: the round_sweep object re-runs the workload in rounds of 256 .. 4096 blocks, which would mix four launch sizes into every kernel's average)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-glue --no-sweep > $OUT/bench_stats.json 2> $OUT/stats.log
cp gpurun_out/bench_detail.json $OUT/bench_stats_detail.json 2>/dev/null   # (stdout carries the short line only: the stage table is in the detail file)
python3 scripts/timeline.py $OUT/stats > $OUT/timeline.txt 2>&1
# the headline workload alone (no secondary / full_rings / sweep objects): every kernel average below is that of one launch shape, directly comparable with
# the HIP-event times of bench.py's roofline object
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-secondary > $OUT/bench_stats_headline.json 2> $OUT/stats_headline.log
cp gpurun_out/bench_detail.json $OUT/bench_stats_headline_detail.json 2>/dev/null
echo "stats pass done: $(tail -c 300 $OUT/stats.log | tr '\n' ' ')"
# The counter passes run the serial schedule: TCC counters are per device, so a side-stream kernel overlapping k_timf2 would be
# charged k_timf2's traffic (seen: 100-200 MB "fetched" by blanker kernels that return at once).  Bytes per kernel do not depend on the schedule.
if [ -z "$STATS_ONLY" ]; then
pmc() {   # name, bench flags
  local wl=$1; shift
  export LRH_PIPELINE=0
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$wl -- python3 bench.py --steps 5 --warmup 2 --no-cpu "$@" > $OUT/bench_fetch_$wl.json 2> $OUT/fetch_$wl.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$wl -- python3 bench.py --steps 5 --warmup 2 --no-cpu "$@" > $OUT/bench_write_$wl.json 2> $OUT/write_$wl.log
  unset LRH_PIPELINE
  echo "pmc passes of $wl done"
}
pmc n1_14_n2_16_n3_12_b8192 --no-secondary
pmc n1_14_n2_12_n3_0_b8192 --fft2-n 12 --fft3-n 0
pmc n1_14_n2_16_n3_12_b8192_full --no-secondary --fft1-float full --fft2-float full     # what the Linrad glue opens (bench.py's full_rings object)
fi
python3 bench.py --steps 50 --warmup 5 > $OUT/bench_plain.json 2> $OUT/plain.log
cp gpurun_out/bench_detail.json $OUT/bench_plain_detail.json 2>/dev/null
# keep the merge small: per-dispatch traces can be large
python3 scripts/summarize_profile.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
