#!/bin/bash
# the drop-in harness at gpu.fft1_batch_n = $@ (default 0 2 4) with the library's per-entry lock-wait / call times (LRH_CALLPROF) and the glue's phase times (HIPSHIM_PROF)
cd "$(dirname "$0")/.." || exit 1
export HIPSHIM_PROF=1 LRH_CALLPROF=1
mkdir -p gpurun_out/glue_in
for n in ${@:-0 2 4}; do
  python3 scripts/glue_trace.py gpurun_out/glue_in "$n" ${KEY:-c2} > gpurun_out/glue_in/cmd$n.txt || exit 1
  for rep in 1 2; do $(cat gpurun_out/glue_in/cmd$n.txt) > gpurun_out/glue_in/cp$n.txt 2> gpurun_out/glue_in/cp$n.err || exit 1; done
done
rm -f gpurun_out/glue_in/in.bin gpurun_out/glue_in/out.bin
