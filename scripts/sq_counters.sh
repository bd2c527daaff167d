#!/bin/bash
# SQ occupancy / stall counters per kernel (two PMC passes), sequential schedule. usage: scripts/sq_counters.sh [bench args]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LRH_PIPELINE=0
OUT=gpurun_out/sq; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 4 --warmup 2 --no-cpu --rounds 1 $@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/b1.json 2> $OUT/l1.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/b2.json 2> $OUT/l2.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p3 -- python3 bench.py $ARGS > $OUT/b3.json 2> $OUT/l3.txt
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for f in glob.glob(sys.argv[1]+"/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void lrh::","")[:24]
        a=acc[k][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,v in acc.items():
    if not any(s in k for s in ("k_fft1","k_timf2","k_fft2","k_blank_scan","k_sumsq","k_sellim")): continue
    print(k)
    for c,(s,n) in sorted(v.items()): print("   %-24s %14.0f  (n=%d)"%(c,s/n,n))
PY
for f in $OUT/l1.txt $OUT/l2.txt $OUT/l3.txt; do tail -n 3 $f; done >&2
find $OUT -name "*.csv" -size +4M -delete
