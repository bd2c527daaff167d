#!/bin/bash
# A/B two builds of the library on one box: scripts/ab.sh [bench args]; variants in ab/lib_*.so
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for v in ab/lib_*.so; do
  cp $v linrad_amd/liblinrad_hip.so
  python bench.py --no-cpu --steps 40 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['stages'].items() if k in ('fft1','timf2','fft2')})"
done; done
