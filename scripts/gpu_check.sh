#!/bin/bash
# one GPU round trip: parity tests, smoke, bench, optional rocprof summary -> gpurun_out/
set -x
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS} 2>/dev/null | tee gpurun_out/bench_last.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('VALUE', d['value'], d['unit'], 'ms/step', d['ms_per_step'])
print('ROOF', d['roofline'])
print('CPU', d['cpu_baseline'])
for k,v in d['stages'].items(): print('  ',k,v)
print(d['blanker'])
"
