"""Copy a rocprofv3 kernel_stats.csv into profiles/ with a markdown summary. usage: save_profile.py <dir> <steps> <tag> <bench_json> [<command>]"""
import glob, json, re, shutil, sys
import pandas as pd
d, steps, tag, bj = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
cmd = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --no-cpu-baseline"
import os
f = max(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
shutil.copy(f, f"profiles/{tag}_kernel_stats_st_pgat_spgnn_3_b512.csv")
shutil.copy(bj, f"profiles/{tag}_bench_under_rocprof.json")
b = json.load(open(bj))
df = pd.read_csv(f).sort_values("TotalDurationNs", ascending=False)
tot = df.TotalDurationNs.sum()
rf, rh = b["roofline"], b.get("roofline_hbm")
if not rf.get("traffic"):                      # the PMC table may be newer than the profiled run
    try:
        rf["traffic"] = json.load(open("profiles/traffic_latest.json")).get("_".join(str(x) for x in [rf["kernel"].replace("spgnn_", "")] + rf["shape"]))
    except Exception:
        pass
if rh and not rh.get("traffic"):
    try:
        rh["traffic"] = json.load(open("profiles/traffic_latest.json")).get("_".join(str(x) for x in [rh["kernel"]] + rh["shape"]))
    except Exception:
        pass
with open(f"profiles/{tag}_kernel_stats_st_pgat_spgnn_3_b512.md", "w") as fp:
    fp.write(f"# rocprofv3 --kernel-trace --stats of bench.py ({tag})\n\n"
             f"Command (MI355X, 1 GPU): `{cmd}`\n\n"
             f"{steps} optimizer steps of st_pgat_spgnn_3, 512 trees (N=76410, E=228206), fp32 parity path (split-fp16 MFMA GEMMs), dropout on: "
             f"{b['warmup']} eager warm-up steps (2 instrumented), 3 capture warm-ups, {b['steps']} timed HIP-graph replays, "
             f"{b.get('eager', {}).get('steps', 0)} eager steps with events around the dominant kernels.  bench.py under the profiler: "
             f"{b['ms_per_step']:.2f} ms/step ({b['value']/1e6:.1f} M layer-edges/s, launch = {b['config']['launch']}); kernel time summed: "
             f"{tot/1e6/steps:.2f} ms/step, {df.Calls.sum()/steps:.0f} launches/step.\n\n"
             f"`roofline` (dominant kernel of the step): `{rf['kernel']}` {rf['shape']}: {rf['avg_launch_ms']*1e3:.1f} us per launch by HIP events "
             f"= {rf['achieved']:.0f} TFLOP/s of executed fp16 MFMA ({rf['frac']:.3f} of 2500), {rf['algorithmic_TFLOPs']:.0f} TFLOP/s as an fp32 product; "
             f"PMC HBM traffic {('%.3f GB' % (rf['traffic'] / 1e9)) if rf.get('traffic') else 'n/a'} per launch.\n"
             + (f"`roofline_hbm` (dominant HBM-bound kernel): `{rh['kernel']}` {rh['shape']}: {rh['avg_launch_ms']*1e3:.1f} us per launch = "
                f"{rh['achieved']:.0f} GB/s ({rh['frac']:.3f} of 8000); PMC traffic {('%.3f GB' % (rh['traffic'] / 1e9)) if rh.get('traffic') else 'n/a'} "
                f"vs {rh['algorithmic_bytes_per_launch']/1e9:.3f} GB algorithmic.\n" if rh else "")
             + "(compare the averages of the same kernels in the table: every launch of the run is in it.)\n\n"
             "| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for _, r in df.head(48).iterrows():
        nm = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:110]
        fp.write(f"| `{nm}` | {r['Calls']} | {r['TotalDurationNs']/1e6:.3f} | {r['AverageNs']/1e3:.1f} | {100*r['TotalDurationNs']/tot:.2f} |\n")
print(open(f"profiles/{tag}_kernel_stats_st_pgat_spgnn_3_b512.md").read()[:2500])
