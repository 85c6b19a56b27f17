#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout -k 10 300 python tools/step_ab.py st_pgat_spgnn_3 2>&1 | tail -2 | tee -a $O/ab.txt
TREES=64 timeout -k 10 300 python tools/step_ab.py st_pgat_spgnn_3 2>&1 | tail -2 | tee -a $O/ab.txt
timeout -k 10 300 python tools/step_ab.py st_gin_3 2>&1 | tail -2 | tee -a $O/ab.txt
