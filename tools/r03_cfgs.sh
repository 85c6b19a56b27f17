#!/bin/bash
# secondary configs, one line each: tools/r03_cfgs.sh <tag>
cd $GRAFT_REPO_ROOT; T=${1:-x}; mkdir -p gpurun_out/r03
for c in st_gcn_3 st_gin_3 st_sage_3 st_gat_3 st_gat_6; do
  python bench.py --config $c --no-cpu-baseline --no-secondary --steps 30 --warmup 8 2>gpurun_out/r03/cfg_${c}_$T.err | grep '^{' | tail -1 > gpurun_out/r03/cfg_${c}_$T.json
  python tools/bench_brief.py gpurun_out/r03/cfg_${c}_$T.json | head -1 | sed "s/^headline/$c/"
done
