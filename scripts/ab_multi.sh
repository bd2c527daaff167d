#!/bin/bash
# scripts/ab_multi.sh "ENV1=a ENV2=b" "ENV1=c" ... -- the headline bench line under several environment settings, alternating, two passes
for i in 1 2; do for e in "$@"; do
  env $e python bench.py --no-cpu --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
f = d['stages'].get('fft2', {})
print('$e', d['value'], d['ms_per_step'], 'fft2', f.get('avg_us'), f.get('avg_us_alone'))"
done; done
