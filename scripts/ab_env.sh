#!/bin/bash
# scripts/ab_env.sh VAR "v0 v1 ..." [reps] [bench args...] -- the same bench line with one environment switch at several values,
# alternating, in one call on one box (run-to-run spread on this pool is 2-3 %: only alternating runs on the same box compare)
VAR=$1; VALS=$2; REPS=${3:-2}; shift 3
for i in $(seq $REPS); do for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], d['ms_per_step'], (d.get('secondary') or {}).get('value'), (d.get('full_rings') or {}).get('value'))"
done; done
