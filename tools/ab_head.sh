#!/bin/bash
# Same-box A/B partner: build ONE kernel source as of a git revision (default HEAD) into build/libchiron_<src>_<rev>.so,
# linked with the working tree's other objects; run it with CHIRON_AMD_LIB=<that .so>.
#   tools/ab_head.sh lstm [rev]
set -e
cd "$(dirname "$0")/../chiron_amd/csrc"
SRC=$1; REV=${2:-HEAD}
mkdir -p ../../build
git show $REV:chiron_amd/csrc/$SRC.hip > ../../build/${SRC}_$REV.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I. -c ../../build/${SRC}_$REV.hip -o ../../build/${SRC}_$REV.o
objs=""
for o in engine gemm stream16 stream32 wino lstm head_ctc beam bn_batch pwl consensus assemble fast5 pipeline; do
  if [ $o = $SRC ]; then objs="$objs ../../build/${SRC}_$REV.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libchiron_${SRC}_$REV.so $objs -lz -ldl
echo build/libchiron_${SRC}_$REV.so
