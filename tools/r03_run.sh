#!/bin/bash
# the GPU suite + the bench line: usage tools/r03_run.sh <tag> [pytest args]
cd $GRAFT_REPO_ROOT
T=${1:-run}; shift
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/r03/pytest_$T.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/r03/pytest_$T.log
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_$T.json 2> gpurun_out/r03/bench_$T.err; echo "bench rc $?"
python tools/bench_brief.py gpurun_out/r03/bench_$T.json
