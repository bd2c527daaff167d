#!/bin/bash
# kernel + copy trace of the drop-in harness at gpu.fft1_batch_n = $1.. (default 0 2 4): what the device does while the glue runs
# usage (on the GPU box): bash scripts/glue_trace.sh [n ...]   -> gpurun_out/glue_prof<n>/, gpurun_out/glue_in/prof<n>.txt (HIPSHIM_PROF lines)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HIPSHIM_PROF=1
mkdir -p gpurun_out/glue_in
for n in ${@:-0 2 4}; do
  python3 scripts/glue_trace.py gpurun_out/glue_in "$n" ${KEY:-c2} > gpurun_out/glue_in/cmd$n.txt || exit 1
  $(cat gpurun_out/glue_in/cmd$n.txt) > gpurun_out/glue_in/plain$n.txt 2> gpurun_out/glue_in/prof$n.txt || exit 1
  rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/glue_prof$n -o g -- $(cat gpurun_out/glue_in/cmd$n.txt) > gpurun_out/glue_in/traced$n.txt 2> gpurun_out/glue_in/traced$n.err || exit 1
done
rm -f gpurun_out/glue_in/in.bin gpurun_out/glue_in/out.bin
