#!/bin/bash
# Little's law check of the drop-in: the same harness run with the timf2 / fft1 rings of the patched Linrad doubled and quadrupled
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/glue_in
for n in ${@:-0 2 4}; do
  python3 scripts/glue_trace.py gpurun_out/glue_in "$n" ${KEY:-c2} > gpurun_out/glue_in/cmd$n.txt || exit 1
  for v in "21 256" "22 256" "22 512" "23 1024" "21 256"; do
    set -- $v
    cmd=$(sed -e "s/timf2pow_log2=[0-9]*/timf2pow_log2=$1/" -e "s/max_fft1n=[0-9]*/max_fft1n=$2/" gpurun_out/glue_in/cmd$n.txt)
    for rep in 1 2 3; do $cmd 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=$n timf2pow_log2=$1 max_fft1n=$2', round(d['samples']/d['loop_seconds']/1e6))"; done
  done
done
rm -f gpurun_out/glue_in/in.bin gpurun_out/glue_in/out.bin
