#!/bin/bash
# The bench's variants on one GPU, one JSON object per round: scripts/variants_round.sh r03   (on the GPU box, via gpurun)
# -> gpurun_out/prof_$R/variants.json (+ rocprofv3 --stats of the two variants the review asked for: --clever, --coupled)
R=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
mkdir -p $OUT/var
run() { name=$1; shift; timeout -k 5 300 python3 bench.py --no-cpu --steps 10 --warmup 3 "$@" > $OUT/var/$name.json 2> $OUT/var/$name.log || echo "$name: rc $?"; }
run headline_sparse --no-secondary
run headline_full_rings --no-secondary --fft1-float full --fft2-float full
run configs1 --fft2-n 12 --fft3-n 0 --no-secondary
run clever --clever --no-secondary
run coupled --coupled --no-secondary
run coupled_stages --coupled --coupled-stages --no-secondary
run spurs8 --spurs 8 --no-secondary --steps 5
run limiter2 --limiter2 --no-secondary
run limiter2_par1_1 --limiter2 --limiter2-par1 1 --no-secondary
run limiter2_par1_0 --limiter2 --limiter2-par1 0 --no-secondary
run streamhost --stream-host --rounds 1 --no-secondary
run realinput --real-input --no-secondary
run n15 --fft1-n 15 --fft2-n 17 --batch 2048 --no-secondary
run n13 --fft1-n 13 --fft2-n 15 --no-secondary
run n12 --fft1-n 12 --fft2-n 14 --no-secondary
run one_round_per_call --rounds 1 --no-secondary --steps 80 --warmup 8   # (a later --steps wins; ten one-round steps would time the pipeline's fill and drain)
for v in clever coupled; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$v -- python3 bench.py --$v --no-cpu --no-secondary --steps 5 --warmup 3 > $OUT/var/${v}_stats.json 2> $OUT/var/${v}_stats.log
  f=$(find $OUT/stats_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -30 "$f" > $OUT/kernel_stats_$v.csv
  find $OUT/stats_$v -name "*kernel_trace.csv" -delete
done
python3 - $OUT <<'PY'
import json, os, sys, glob
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "var", "*.json"))):
    name = os.path.basename(f)[:-5]
    if name.endswith("_stats"):
        continue
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        out[name] = {"error": repr(e)}
        continue
    out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"][:160],
                 "stages_us": d.get("stage_us") or {k: [v["avg_us"], v.get("avg_us_alone")] for k, v in d.get("stages", {}).items()},   # (the short line carries avg_us per stage)
                 "blanker": d.get("blanker"), "spurs": d.get("spurs"), "routing": d.get("routing")}
json.dump(out, open(os.path.join(sys.argv[1], "variants.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v.get("value"), v.get("ms_per_step"))
PY
