set -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 5 300 python bench.py --no-cpu --steps 5 --warmup 2 "$@" > gpurun_out/var_$name.json 2> gpurun_out/var_$name.log; rc=$?; python3 -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/var_$name.json').read().strip().splitlines()[-1]); print('$name', 'rc', $rc, round(d['value'],1), d['unit'], d['n_gpus'], d['config'].get('parallelism'))
except Exception as e: print('$name', 'rc', $rc, 'NO JSON', e)
"; }
run limiter2 --limiter2 --no-secondary
run streamhost --stream-host --rounds 1 --no-secondary
run realinput --real-input --no-secondary
run n15 --fft1-n 15 --fft2-n 17 --batch 2048 --no-secondary --no-sellim
LRH_BENCH_BACKEND=gloo LRH_BENCH_SAME_DEVICE=1 run coupled2 --gpus 2 --coupled --no-secondary
LRH_BENCH_BACKEND=gloo LRH_BENCH_SAME_DEVICE=1 run combine4 --gpus 4 --no-secondary
