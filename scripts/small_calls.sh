#!/bin/bash
# throughput of small lrh_wideband_dsp calls (fft1 blocks per call, one round per call) on the default workload, per schedule
out=gpurun_out/small_calls.txt; : > $out
for pl in ${PIPELINES:-default 2 0}; do for b in ${BATCHES:-1 4 16 64 256}; do
  if [ "$pl" = default ]; then unset LRH_PIPELINE; else export LRH_PIPELINE=$pl; fi
  timeout -k 5 200 python bench.py --no-cpu --no-secondary --fft1-float full --fft2-n ${FFT2N:-16} --batch $b --rounds 1 --steps ${STEPS:-2000} --warmup 100 > gpurun_out/_s.json 2> gpurun_out/_s.log || { echo "pipeline $pl batch $b FAILED" >> $out; continue; }
  python3 - $pl $b >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_s.json").read().strip().splitlines()[-1])
print("LRH_PIPELINE", sys.argv[1], "blocks/call", sys.argv[2], "Msamples/s", round(d["value"], 1), "us/call", round(1000 * d["ms_per_step"], 1))
PY
done; done
cat $out
# the same without the selective limiter (its one-workgroup kernel, once per fft1 averaging period, is half of a small call's device time)
if [ -n "$NOSELLIM" ]; then
  unset LRH_PIPELINE
  for b in ${BATCHES:-1 4 16 64 256}; do
    timeout -k 5 200 python bench.py --no-cpu --no-secondary --no-sellim --fft1-float full --fft2-n ${FFT2N:-16} --batch $b --rounds 1 --steps ${STEPS:-2000} --warmup 100 > gpurun_out/_s.json 2> gpurun_out/_s.log || continue
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/_s.json').read().strip().splitlines()[-1])
print('no limiter, blocks/call', $b, 'Msamples/s', round(d['value'],1), 'us/call', round(1000*d['ms_per_step'],1))" >> $out
  done
  tail -5 $out
fi
