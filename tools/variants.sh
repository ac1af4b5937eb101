#!/bin/bash
# Build variants of ONE kernel source into build/ (never over the product library) and link each with the tree's other objects:
#   tools/variants.sh lstm CHIRON_W32_VARIANT 1 2 3          ->  build/libchiron_lstm_CHIRON_W32_VARIANT_1.so ...
#     TIMING builds (csrc/timing_variants.h: parts of a kernel switched off, results are garbage): compiled with
#     -DCHIRON_TIMING_BUILD; load them with CHIRON_AMD_LIB=<that .so> CHIRON_ALLOW_TIMING_BUILD=1 (chiron_amd/_lib.py refuses otherwise).
#   tools/variants.sh --product lstm CHIRON_GATE_MATH 1 2    ->  build/libchiron_lstm_CHIRON_GATE_MATH_1.so ...
#     alternative PRODUCT forms that compute correct results (gate math accuracy levels, ...): CHIRON_AMD_LIB=<that .so> alone.
set -e
cd "$(dirname "$0")/../chiron_amd/csrc"
TIMING=-DCHIRON_TIMING_BUILD
if [ "$1" = "--product" ]; then TIMING=""; shift; fi
SRC=$1; MACRO=$2; shift 2
mkdir -p ../../build
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $TIMING -D$MACRO=$v -c $SRC.hip -o ../../build/${SRC}_${MACRO}_$v.o
  objs=""
  for o in engine gemm stream16 stream32 wino lstm head_ctc beam bn_batch pwl consensus assemble fast5 pipeline; do
    if [ $o = $SRC ]; then objs="$objs ../../build/${SRC}_${MACRO}_$v.o"; else objs="$objs $o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libchiron_${SRC}_${MACRO}_$v.so $objs -lz -ldl
done
