"""Aggregate a rocprofv3 kernel_stats.csv by category. usage: prof_summary.py <dir> <steps>"""
import glob, re, sys
import pandas as pd
d, steps = sys.argv[1], int(sys.argv[2])
import os
f = max(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
df = pd.read_csv(f)
def cat(n):
    if n.startswith("Cijk"): return "rocblas_gemm"
    if "gat_" in n or "spmm" in n or "sgd_momentum" in n or "scores_" in n or "head_mean" in n:
        return "spgnn:" + re.sub(r"<.*|\(.*", "", n.split("::")[-1])
    if "CatArray" in n: return "torch:cat"
    if "dropout" in n or "masked_scale" in n: return "torch:dropout"
    if "reduce_kernel" in n: return "torch:reduce(sum/mean)"
    if "elementwise" in n: return "torch:elementwise"
    if "copyBuffer" in n or "fillBuffer" in n: return "rocclr:copy/fill"
    return "other:" + n[:50]
df["cat"] = df["Name"].map(cat)
g = df.groupby("cat").agg(calls=("Calls", "sum"), total_ms=("TotalDurationNs", lambda x: x.sum() / 1e6)).sort_values("total_ms", ascending=False)
g["ms_per_step"] = g["total_ms"] / steps
g["calls_per_step"] = g["calls"] / steps
print(g.to_string())
print("total ms/step", g["total_ms"].sum() / steps, "launches/step", g["calls"].sum() / steps)
top = df.sort_values("TotalDurationNs", ascending=False).head(14)
for _, r in top.iterrows():
    nm = r["Name"]
    m = re.search(r"(Cijk_\w+?)_S_B.*?(MT\d+x\d+x\d+)", nm)
    nm = f"GEMM {m.group(1)} {m.group(2)}" if m else re.sub(r"\(anonymous namespace\)::|void ", "", nm)[:70]
    print(f"{nm:72s} calls {r['Calls']:4d} avg {r['AverageNs']/1e3:8.1f} us  {r['Percentage']:.2f}%")
