#!/bin/bash
# 64-tree headline step under the kernel trace + per-step breakdown
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03
bash $R/tools/r03_prof.sh t64 --trees 64 && cd $R && python3 tools/trace_steps.py gpurun_out/r03/prof_t64 > gpurun_out/r03/t64_steps.txt 2>&1
tail -5 gpurun_out/r03/t64_steps.txt
