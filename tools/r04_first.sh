#!/bin/bash
# round 4, first GPU call: new tests first, then the whole GPU suite, the batch-cycle leg and the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_arena.py tests/test_hip_gemm.py -x -q -m gpu > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/new_tests.log
timeout -k 10 300 python bench.py --batch-cycle-only > $O/batch_cycle.json 2> $O/batch_cycle.err; echo "batch cycle rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/all_tests.log 2>&1; echo "all tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/all_tests.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<P
import json
try:
    d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "gemm", d.get("gemm", {}).get("ms_per_step"), "k123", d.get("roofline_k123", {}).get("ms_per_step"), "frac", d["roofline"].get("executed_mfma_frac"), "hbm", d["roofline"].get("hbm", {}).get("frac"))
    print(json.dumps(d["config"].get("batch_cycle_64")))
    for k, v in d.get("secondary", {}).items():
        print(k, v.get("ms_per_step"), v.get("error"))
except Exception as e:
    print("ERR", e)
try:
    b = json.loads(open("$O/batch_cycle.json").read().strip().splitlines()[-1])["batch_cycle_64"]
    print({k: v for k, v in b.items() if k != "batches"})
    for q in b["batches"]: print(q)
except Exception as e:
    print("ERR", e)
P
