#!/bin/bash
# End-of-round measurement on one GPU box: GPU tests, bench.py under rocprofv3 (kernel trace), PMC traffic of the
# GEMM kernels, the plain bench line, and the other configurations.  Everything lands under gpurun_out/final/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2> $O/bench_under_rocprof.err)
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
bash tools/pmc_gemm_traffic.sh > $O/pmc_gemm_traffic.log 2>&1
python bench.py > $O/bench.log 2> $O/bench.err; grep '^{' $O/bench.log | tail -1 > $O/bench.json
python bench.py --eager --no-cpu-baseline > $O/bench_eager.log 2>&1
for c in st_gat_3 st_gat_6 st_gcn_3 st_gin_3 st_sage_3; do
  python bench.py --config $c --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | grep '^{' | tail -1 > $O/cfg_$c.json
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/final"
for f in ["bench.json", "bench_under_rocprof.json"] + sorted(glob.glob(O + "/cfg_*.json")):
    f = f if f.startswith("/") else O + "/" + f
    try:
        d = json.load(open(f))
        print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"] / 1e6, 1), d["config"]["launch"], (d.get("eager") or {}).get("ms_per_step"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
