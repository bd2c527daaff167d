#!/bin/bash
# oracle/build_shim_harness.sh -- TEST INFRASTRUCTURE, build container only (needs the reference tree).
#
# Builds, from the reference objects as integration/linrad_hip.patch leaves them + the head-less driver oracle/ref_harness.c:
#
#   oracle/_ref/shim_harness      integration/hipshim.c compiled over the ORACLE's C ABI (oracle/shim_alias.h: lrh_* -> lro_*), so the
#                                 Linrad-side glue executes without a GPU (tests/test_shim_exec_cpu.py)
#   oracle/_ref/shim_harness_hip  the same objects, the same driver, hipshim.c compiled AS SHIPPED and linked to
#                                 linrad_amd/liblinrad_hip.so -- what a patched xlinrad64 executes: patched call sites -> hipshim.c ->
#                                 the HIP library with lrh_timf1_write_async, lrh_host_register and the pinned read-backs.  Runs on the GPU
#                                 box only (tests/test_gpu_shim.py); it travels there like ref_harness (oracle/_ref is not gpurun-ignored)
#                                 and finds the library through $ORIGIN.
#
# fft1_b case 21 -> fft1_c -> make_timf2 -> first_noise_blanker -> make_fft2 -> fft2_mix1_* (and the second-fft-off chain
# fft1_c -> fft1_mix1_*) run through the reference's own, patched, call sites.
#
# The patched copies of the reference sources live in a scratch directory under /tmp for the duration of the build; only
# objects and the binaries are written, and only into oracle/_ref/ (git-ignored).
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
[ -d "$REF" ] || { echo "no reference tree at $REF"; exit 2; }
[ -f "$ROOT/linrad_amd/liblinrad_hip.so" ] || { echo "build linrad_amd/liblinrad_hip.so first (lrh_config_defaults)"; exit 2; }
make -C "$HERE" oracle >/dev/null
T=$(mktemp -d /tmp/linrad_shim.XXXXXX)
trap 'rm -rf "$T"' EXIT
OUT="$HERE/_ref/shim"
mkdir -p "$OUT"
cp "$REF"/*.h "$T"/
TOUCHED="fft1var.c buf.c wcw.c fft1.c timf2.c blank1.c fft2.c mix1.c sellim.c rxin.c spursub.c spur.c"
for f in $TOUCHED; do cp "$REF/$f" "$T/"; done
(cd "$T" && patch -s -p1 --no-backup-if-mismatch < "$ROOT/integration/linrad_hip.patch")
cp "$ROOT/integration/hipshim.c" "$ROOT/integration/hipshim.h" "$T"/
DEFS="-DOSNUM=1 -DCPU=1 -DIA64=1 -DHAVE_OSS=0 -DHAVE_X11=0 -DHAVE_SHM=0 -DHAVE_SVGALIB=0 -DSERVER=0 -DOPENCL_PRESENT=0 -DHAVE_CUFFT=0 -DOSSD=0"
CFLAGS="-O2 -ffast-math -fomit-frame-pointer -w $DEFS"
# the object list of oracle/Makefile (REFSRC); patched where the patch touches a file, every file against the patched headers
SRC="fft0 fft1 fft1_re timf2 blank1 fft2 mix1 fft3 mix2 wide_graph wcw buf sellim spur spursub llsq fft1var fft2var fft3var uivar screenvar sigvar selvar blnkvar calvar thrvar"
OBJ=""
for s in $SRC; do
  if [ -f "$T/$s.c" ]; then src="$T/$s.c"; else src="$REF/$s.c"; fi
  gcc $CFLAGS -I"$T" -I"$ROOT/include" -c "$src" -o "$OUT/$s.o"
  OBJ="$OBJ $OUT/$s.o"
done
# (1) the glue over the oracle's ABI
gcc -O2 -Wall -Wno-unused-parameter $DEFS -I"$T" -I"$HERE" -I"$ROOT/include" -include "$HERE/shim_alias.h" -c "$T/hipshim.c" -o "$OUT/hipshim.o"
gcc -O1 -w -no-pie $DEFS -DSHIM_HARNESS=1 -I"$T" -I"$HERE" -I"$ROOT/include" -include "$HERE/shim_alias.h" "$HERE/ref_harness.c" $OBJ "$OUT/hipshim.o" \
    -o "$HERE/_ref/shim_harness" -L"$HERE" -llinrad_oracle -L"$ROOT/linrad_amd" -llinrad_hip \
    -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../../linrad_amd' -lm -lpthread -Wl,--unresolved-symbols=ignore-all
echo "built $HERE/_ref/shim_harness"
# SAN=1: the same binary with AddressSanitizer + UBSan on the glue and the driver (CPU only: GPU sanitizers are not available on this pool)
if [ -n "$SAN" ]; then
  gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -Wno-unused-parameter $DEFS -I"$T" -I"$HERE" -I"$ROOT/include" -include "$HERE/shim_alias.h" -c "$T/hipshim.c" -o "$OUT/hipshim_asan.o"
  gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -w -no-pie $DEFS -DSHIM_HARNESS=1 -I"$T" -I"$HERE" -I"$ROOT/include" -include "$HERE/shim_alias.h" "$HERE/ref_harness.c" $OBJ "$OUT/hipshim_asan.o" \
      -o "$HERE/_ref/shim_harness_asan" -L"$HERE" -llinrad_oracle -L"$ROOT/linrad_amd" -llinrad_hip \
      -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../../linrad_amd' -lm -lpthread -Wl,--unresolved-symbols=ignore-all
  echo "built $HERE/_ref/shim_harness_asan"
  # ... and with ThreadSanitizer (glue + driver; the reference objects are not instrumented): the stage-thread modes of the harness
  gcc -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Wall -Wno-unused-parameter $DEFS -I"$T" -I"$HERE" -I"$ROOT/include" -include "$HERE/shim_alias.h" -c "$T/hipshim.c" -o "$OUT/hipshim_tsan.o"
  gcc -O1 -g -fsanitize=thread -fno-omit-frame-pointer -w -no-pie $DEFS -DSHIM_HARNESS=1 -I"$T" -I"$HERE" -I"$ROOT/include" -include "$HERE/shim_alias.h" "$HERE/ref_harness.c" $OBJ "$OUT/hipshim_tsan.o" \
      -o "$HERE/_ref/shim_harness_tsan" -L"$HERE" -llinrad_oracle -L"$ROOT/linrad_amd" -llinrad_hip \
      -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,'$ORIGIN/../../linrad_amd' -lm -lpthread -Wl,--unresolved-symbols=ignore-all
  echo "built $HERE/_ref/shim_harness_tsan"
fi
# (2) the glue as shipped, over liblinrad_hip.so (no alias header: lrh_* are the library's own entry points)
gcc -O2 -Wall -Wno-unused-parameter $DEFS -I"$T" -I"$ROOT/include" -c "$T/hipshim.c" -o "$OUT/hipshim_hip.o"
gcc -O1 -w -no-pie $DEFS -DSHIM_HARNESS=1 -I"$T" -I"$HERE" -I"$ROOT/include" "$HERE/ref_harness.c" $OBJ "$OUT/hipshim_hip.o" \
    -o "$HERE/_ref/shim_harness_hip" -L"$ROOT/linrad_amd" -llinrad_hip \
    -Wl,-rpath,'$ORIGIN/../../linrad_amd' -lm -lpthread -Wl,--unresolved-symbols=ignore-all
if nm -D --undefined-only "$HERE/_ref/shim_harness_hip" | grep -q ' lro_'; then echo "shim_harness_hip references the oracle"; exit 3; fi
echo "built $HERE/_ref/shim_harness_hip"
