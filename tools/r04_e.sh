#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_models.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log
timeout -k 10 300 python tools/tn_tiles.py 2>&1 | tee $O/tn_tiles_512.txt
timeout -k 10 300 python tools/tn_tiles.py 9641 2>&1 | tee $O/tn_tiles_64.txt
for t in 512 64; do
  timeout -k 10 300 python tools/step_toggle_ab.py $t base= tn128=ops.TN_TILE:128 tn256=ops.TN_TILE:256 2>&1 | tail -1 | tee -a $O/ab.txt
done
