#!/bin/bash
# A/B of an environment switch on the bench: scripts/ab_env.sh VAR "v1 v2 ..." [bench flags]
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout -k 10 240 python bench.py --no-cpu --no-secondary --steps 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], {k:(v['avg_us'],v.get('avg_us_alone')) for k,v in d['stages'].items() if k in ('fft1w','timf2s','clever','blanker','fft2','spur')})" || break
done
