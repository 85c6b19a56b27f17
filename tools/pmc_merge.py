"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_step.py.

usage: pmc_merge.py <fetch_dir> <write_dir> <manifest.json> <out.md> [<table.json to update>]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The LAST len(manifest) profiler rows
whose kernel names carry one of the stems are the instrumented step; they are matched to bench.py's (kernel, shape) keys
in launch order and the stem is checked row by row."""
import glob, json, re, sys
import pandas as pd

STEM_RE = re.compile(r"\b(gat_fwd|gat_bwd_dst|gat_bwd_src|gat_agg_fwd|gat_agg_bwd_dst|gat_agg_bwd_src|lspe_fwd|lspe_bwd_dst|lspe_bwd_src|gemm_nt_pair|gemm_tn_pair|gemm_nt|gemm_tn)")


def stem_of(name):
    m = STEM_RE.search(name)
    return m.group(1) if m else None


def counter_rows(d, counter, n):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df["Counter_Name"] == counter]
    df = df.groupby(["Dispatch_Id", "Kernel_Name"], as_index=False)["Counter_Value"].sum().sort_values("Dispatch_Id")
    df["stem"] = df["Kernel_Name"].map(stem_of)
    df = df[df["stem"].notna()]
    return df.tail(n).reset_index(drop=True)


manifest = json.load(open(sys.argv[3]))
fetch, write = counter_rows(sys.argv[1], "FETCH_SIZE", len(manifest)), counter_rows(sys.argv[2], "WRITE_SIZE", len(manifest))
assert len(fetch) == len(write) == len(manifest), (len(fetch), len(write), len(manifest))
acc = {}
for i, (stem, key) in enumerate(manifest):
    assert fetch.loc[i, "stem"] == stem and write.loc[i, "stem"] == stem, (i, stem, fetch.loc[i, "Kernel_Name"], write.loc[i, "Kernel_Name"])
    b = (2.0 * fetch.loc[i, "Counter_Value"] + write.loc[i, "Counter_Value"]) * 1024.0
    acc.setdefault(key, []).append((b, fetch.loc[i, "Counter_Value"], write.loc[i, "Counter_Value"], fetch.loc[i, "Kernel_Name"]))
out = {k: sum(x[0] for x in v) / len(v) for k, v in acc.items()}
with open(sys.argv[4], "w") as fp:
    fp.write("HBM traffic per launch: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over one eagerly issued "
             "training step (tools/pmc_step.py); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction of the guide).\n\n"
             "| bench key | kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch |\n|---|---|---|---|---|---|\n")
    for k, v in acc.items():
        n = len(v)
        nm = re.sub(r"\(anonymous namespace\)::|void ", "", v[0][3]).split("(")[0][:60]
        fp.write(f"| `{k}` | `{nm}` | {n} | {sum(x[1] for x in v)/n:.0f} | {sum(x[2] for x in v)/n:.0f} | {out[k]/1e9:.4f} GB |\n")
if len(sys.argv) > 5:
    try:
        tab = json.load(open(sys.argv[5]))
    except Exception:
        tab = {}
    tab.update(out)
    json.dump(tab, open(sys.argv[5], "w"), indent=1, sort_keys=True)
print(len(out), "keys;", sum(len(v) for v in acc.values()), "launches")
