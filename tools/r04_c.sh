#!/bin/bash
# new parity-at-size tests, the two-process loop with the GEMM-only worker, and the N = 2 rehearsal of bench.py (gloo on one GPU)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_parity_at_size.py -x -q -m gpu -s > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
grep -E "trees|passed|failed|Error" $O/parity.log | tail -20
timeout -k 10 900 python -m pytest tests/test_two_process.py -x -q -m gpu > $O/twoproc.log 2>&1; echo "two-process rc=$?" | tee -a $O/summary.txt
tail -3 $O/twoproc.log
SPGNN_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 3 --trees 64 --no-kernel-timers > $O/rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc=$?" | tee -a $O/summary.txt
python - <<P
import json
try:
    d = json.loads([l for l in open("$O/rehearsal.json") if l.startswith("{")][-1])
    print("n_gpus", d["n_gpus"], "ms", d["ms_per_step"], "launch", d["config"]["launch"], "comm", json.dumps(d.get("comm")))
except Exception as e:
    print("ERR", e); print(open("$O/rehearsal.err").read()[-1500:])
P
