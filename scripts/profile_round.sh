#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats + PMC (FETCH_SIZE / WRITE_SIZE in separate passes).
# usage (on the GPU box, via gpurun): scripts/profile_round.sh r01
R=${1:-r01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
mkdir -p $OUT
ARGS="--steps 10 --warmup 3 --no-cpu ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
python3 bench.py --steps 50 --warmup 5 ${BENCH_ARGS} > $OUT/bench_plain.json 2> $OUT/plain.log
find $OUT -name "*.csv" | head -20
# keep the merge small: per-dispatch traces can be large
find $OUT -name "*kernel_trace.csv" -size +8M -delete
python3 scripts/summarize_profile.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
