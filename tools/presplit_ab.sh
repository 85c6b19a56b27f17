cd $GRAFT_REPO_ROOT
for r in 1 2; do
python -c "import sys; sys.argv=['bench.py','--no-cpu-baseline']; import spgnn_amd.ops as o; o.PRESPLIT_B=False; import runpy; runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | grep '^{' | tail -1 > gpurun_out/ab_off_$r.json
python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/ab_on_$r.json
done
python - <<'P'
import json
for n in ("off_1","on_1","off_2","on_2"):
    d=json.load(open(f"gpurun_out/ab_{n}.json")); g=d["gemm"]["kernels"]; r=d["roofline"]
    print(n, round(d["ms_per_step"],3), "nt ms", round(g["gemm_nt"]["ms_per_step"],3), "tn", round(g["gemm_tn"]["ms_per_step"],3), "exec frac", round(r["executed_mfma_frac"],4))
P
