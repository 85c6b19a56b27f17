#!/bin/bash
# HBM bytes per launch of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over
# tools/gemm_driver.py; result merged into profiles/traffic_latest.json under bench.py's gemm keys.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm_traffic; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/gemm_driver.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/gemm_driver.py > $O/write.log 2>&1
cd $R && python3 - <<'P'
import glob, json, pandas as pd
O = "gpurun_out/pmc_gemm_traffic"
def rows(d, c):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f); df = df[(df.Counter_Name == c) & df.Kernel_Name.str.contains("gemm_nt_f16x3|gemm_tn_f16x3")]
    return df.groupby(["Dispatch_Id", "Kernel_Name", "Grid_Size"], as_index=False).Counter_Value.sum().sort_values("Dispatch_Id").reset_index(drop=True)
fe, wr = rows(O + "/fetch", "FETCH_SIZE"), rows(O + "/write", "WRITE_SIZE")
assert len(fe) == len(wr)
# gemm_driver.py launch order per shape (K, C): nt(x, w) then tn(g, x), twice each; shapes (1063,1024), (768,512), (192,4096)
keys = []
for (K, C) in [(1063, 1024), (768, 512), (192, 4096), (384, 1024)]:
    keys += [f"gemm_nt_76410_{C}_{K}", f"gemm_tn_76410_{C}_{K}"] * 2
out = {}
with open(O + "/summary.md", "w") as fp:
    fp.write("| bench key | kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch = (2 FETCH + WRITE) KiB |\n|---|---|---|---|---|\n")
    for i, k in enumerate(keys):
        b = (2.0 * fe.Counter_Value[i] + wr.Counter_Value[i]) * 1024.0
        out.setdefault(k, []).append(b)
        fp.write(f"| `{k}` | `{fe.Kernel_Name[i][:40]}` | {fe.Counter_Value[i]:.0f} | {wr.Counter_Value[i]:.0f} | {b/1e9:.3f} GB |\n")
out = {k: sum(v) / len(v) for k, v in out.items()}
json.dump(out, open(O + "/gemm_traffic.json", "w"), indent=1)
print(open(O + "/summary.md").read())
P
