#!/bin/bash
# range monitor + deferred attention-vector gradients: tests, then A/B of the step (DEFER on / off) for fp32 and bf16
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_models.py tests/test_hip_lspe.py tests/test_hip_bf16.py tests/test_arena.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests.log
for t in 512 64; do
  timeout -k 10 300 python tools/step_toggle_ab.py $t base= nodefer=ops.DEFER_ATTN_GRADS:0 2>&1 | tail -1 | tee -a $O/ab.txt
done
CONFIG=st_gat_6 DTYPE=bf16 timeout -k 10 300 python tools/step_toggle_ab.py 512 base= nodefer=ops.DEFER_ATTN_GRADS:0 2>&1 | tail -1 | tee -a $O/ab.txt
CONFIG=st_gat_3 timeout -k 10 300 python tools/step_toggle_ab.py 64 base= nodefer=ops.DEFER_ATTN_GRADS:0 2>&1 | tail -1 | tee -a $O/ab.txt
