#!/bin/bash
# Build instrumented variants of ONE kernel source into build/ (never the product library) for timing experiments:
#   tools/variants.sh lstm CHIRON_W32_VARIANT 1 2 3   ->  build/libchiron_lstm_CHIRON_W32_VARIANT_1.so ...
# and run them with CHIRON_AMD_LIB=<that .so> (chiron_amd/_lib.py).
set -e
cd "$(dirname "$0")/../chiron_amd/csrc"
SRC=$1; MACRO=$2; shift 2
mkdir -p ../../build
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -D$MACRO=$v -c $SRC.hip -o ../../build/${SRC}_${MACRO}_$v.o
  objs=""
  for o in engine gemm stream16 stream32 wino lstm head_ctc beam bn_batch pwl consensus assemble fast5; do
    if [ $o = $SRC ]; then objs="$objs ../../build/${SRC}_${MACRO}_$v.o"; else objs="$objs $o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libchiron_${SRC}_${MACRO}_$v.so $objs -lz -ldl
done
