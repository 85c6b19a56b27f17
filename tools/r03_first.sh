#!/bin/bash
# first GPU call of round 3: the GPU suite, the bench line with its secondary legs, GEMM with / without the SLP vectorizer
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r03/pytest_gpu.log
tail -3 gpurun_out/r03/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc $?"
timeout -k 10 200 python tools/gemm_ab.py nt > gpurun_out/r03/gemm_ab_nt.log 2>&1; echo "ab nt rc $?"
timeout -k 10 200 python tools/gemm_ab.py tn > gpurun_out/r03/gemm_ab_tn.log 2>&1; echo "ab tn rc $?"
cat gpurun_out/r03/gemm_ab_nt.log gpurun_out/r03/gemm_ab_tn.log
