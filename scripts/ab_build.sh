#!/bin/bash
# Tuning aid: build variants of libhgs.so that differ in -D flags of ONE source file.
#   scripts/ab_build.sh render.hip  name1 "-DX=0"  name2 "-DX=1 -DY=2" ...
# -> ab_variants/libhgs_<name>.so ; on the GPU box: scripts/ab_run.sh name1 name2 ...
set -e
SRC=$1; shift
C=hierarchical-3d-gaussians_amd/csrc
make -s -C $C
mkdir -p ab_variants
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $( [ "$SRC" = render.hip ] && echo -fno-slp-vectorize ) -I$C $flags -x hip -c $C/$SRC -o ab_variants/$SRC.$name.o
  objs=$(ls build/hgs/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_variants/libhgs_$name.so $objs ab_variants/$SRC.$name.o
  echo "built ab_variants/libhgs_$name.so ($flags)"
done
