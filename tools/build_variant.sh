#!/bin/bash
# usage: tools/build_variant.sh <name> <kernels|lspe|graph|gemm|bf16> [-DSWITCH=v ...]   ->  build/variants/<name>.so
# (the named translation unit rebuilt with the switches, the other objects taken from build/obj)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; unit=$2; shift; shift
extra=""; if [ $unit == kernels ] || [ $unit == lspe ]; then extra="-fno-slp-vectorize -fno-vectorize -DSPGNN_NO_SLP_VECTORIZE"; fi      # as csrc/build.py
mkdir -p $R/build/variants $R/build/vobj
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I $R/include -I $R/spgnn_amd/csrc $extra "$@" -c $R/spgnn_amd/csrc/spgnn_$unit.hip -o $R/build/vobj/$name.o
objs=""
for u in kernels lspe graph gemm bf16; do
  if [ $u == $unit ]; then objs="$objs $R/build/vobj/$name.o"; else objs="$objs $R/build/obj/spgnn_$u.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/$name.so $objs
echo built $R/build/variants/$name.so
