#!/bin/bash
# like glue_trace.sh with the HIP runtime API calls of every thread in the trace too (which call the host spends its time in)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HIPSHIM_PROF=1
mkdir -p gpurun_out/glue_in
for n in ${@:-0 4}; do
  python3 scripts/glue_trace.py gpurun_out/glue_in "$n" ${KEY:-c2} > gpurun_out/glue_in/cmd$n.txt || exit 1
  rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d gpurun_out/glue_api$n -o g -- $(cat gpurun_out/glue_in/cmd$n.txt) > gpurun_out/glue_in/api$n.txt 2> gpurun_out/glue_in/api$n.err || exit 1
done
rm -f gpurun_out/glue_in/in.bin gpurun_out/glue_in/out.bin
