#!/bin/bash
# kernel trace + per-step breakdown of one config: tools/r03_cfgprof.sh <config> [bench args]
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03
C=$1; shift
bash $R/tools/r03_prof.sh $C --config $C "$@" && cd $R && python3 tools/trace_steps.py gpurun_out/r03/prof_$C > gpurun_out/r03/steps_$C.txt 2>&1
tail -3 gpurun_out/r03/steps_$C.txt
