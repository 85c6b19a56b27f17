#!/bin/bash
# usage: tools/build_variant.sh <name> [-DSWITCH=v ...]   ->  build/variants/<name>.so  (kernels file rebuilt with the switches,
# the other objects taken from build/obj)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/variants $R/build/vobj
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I $R/include -I $R/spgnn_amd/csrc "$@" -c $R/spgnn_amd/csrc/spgnn_kernels.hip -o $R/build/vobj/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/$name.so $R/build/vobj/$name.o $R/build/obj/spgnn_gemm.hip.o $R/build/obj/spgnn_bf16.hip.o
echo built $R/build/variants/$name.so
