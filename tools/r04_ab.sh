#!/bin/bash
# a_presplit on / off in one process (HIP-graph replays, interleaved rounds)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
for t in 512 64; do
  timeout -k 10 300 python tools/step_toggle_ab.py $t base= aps_off=ops.A_PRESPLIT:0 2>&1 | tail -1 | tee -a $O/ab.txt
done
CONFIG=st_gat_3 timeout -k 10 300 python tools/step_toggle_ab.py 512 base= aps_off=ops.A_PRESPLIT:0 2>&1 | tail -1 | tee -a $O/ab.txt
CONFIG=st_sage_3 timeout -k 10 300 python tools/step_toggle_ab.py 512 base= aps_off=ops.A_PRESPLIT:0 2>&1 | tail -1 | tee -a $O/ab.txt
