#!/bin/bash
# Applies integration/linrad_hip.patch to a scratch copy of the Linrad sources it touches and compiles every touched
# object plus hipshim.c with the head-less defines oracle/Makefile uses for the reference (-O2 -ffast-math, Makefile.in:104-111).
# Build container only (needs the reference tree); nothing is written outside the scratch directory.
#   usage: integration/check_patch.sh [reference dir]     exit 0 = patch applies and everything compiles
set -e
REF=${1:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
[ -d "$REF" ] || { echo "no reference tree at $REF"; exit 2; }
T=$(mktemp -d /tmp/linrad_patch.XXXXXX)
trap 'rm -rf "$T"' EXIT
cp "$REF"/*.h "$T"/
TOUCHED="fft1var.c buf.c wcw.c fft1.c timf2.c blank1.c fft2.c mix1.c sellim.c rxin.c spursub.c spur.c"
for f in $TOUCHED; do cp "$REF/$f" "$T/"; done
(cd "$T" && patch -p1 --no-backup-if-mismatch < "$HERE/linrad_hip.patch")
cp "$HERE/hipshim.c" "$HERE/hipshim.h" "$T"/
DEFS="-DOSNUM=1 -DCPU=1 -DIA64=1 -DHAVE_OSS=0 -DHAVE_X11=0 -DHAVE_SHM=0 -DHAVE_SVGALIB=0 -DSERVER=0 -DOPENCL_PRESENT=0 -DHAVE_CUFFT=0 -DOSSD=0"
for f in $TOUCHED hipshim.c; do
  gcc -O2 -ffast-math -w $DEFS -I"$T" -I"$ROOT/include" -c "$T/$f" -o "$T/${f%.c}.o"
  echo "compiled $f"
done
# hipshim.c once more with warnings on: it is our code
gcc -O2 -Wall -Wextra -Wno-unused-parameter $DEFS -I"$T" -I"$ROOT/include" -c "$T/hipshim.c" -o "$T/hipshim_w.o"
# every hook the patch adds resolves to a function hipshim.c defines
need=$(grep -ho 'hip_[a-z0-9_]*(' "$HERE/linrad_hip.patch" | sort -u | tr -d '(')
have=$(nm "$T/hipshim.o" | awk '$2=="T"{print $3}')
for s in $need; do echo "$have" | grep -qx "$s" || { echo "hook $s is not defined by hipshim.c"; exit 1; }; done
# and the library exports what hipshim.c calls
if [ -f "$ROOT/linrad_amd/liblinrad_hip.so" ]; then
  for s in $(nm "$T/hipshim.o" | awk '$1=="U" && $2 ~ /^lrh_/{print $2}'); do
    nm -D "$ROOT/linrad_amd/liblinrad_hip.so" | grep -q " T $s$" || { echo "liblinrad_hip.so lacks $s"; exit 1; }
  done
fi
grep -c "GPU_HIP" "$T/fft1.c" "$T/wcw.c" "$T/globdef.h" | tr '\n' ' '; echo
echo "patch ok: applies to $REF and all touched objects compile"
