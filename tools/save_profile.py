"""Copy a rocprofv3 kernel_stats.csv into profiles/ with a markdown summary.
usage: save_profile.py <rocprof dir> <tag> <bench_json> <command> [suffix]"""
import glob, json, os, re, shutil, sys
import pandas as pd

d, tag, bj, cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
sfx = sys.argv[5] if len(sys.argv) > 5 else ""
f = max(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
b = json.load(open(bj))
cfgname, dtype = b["config"]["workload"].split(",")[0].split()[0], b["dtype"]
stem = f"profiles/{tag}_kernel_stats_{cfgname}_{dtype}{sfx}"
shutil.copy(f, stem + ".csv")
shutil.copy(bj, f"profiles/{tag}_bench_under_rocprof_{cfgname}_{dtype}{sfx}.json")
df = pd.read_csv(f).sort_values("TotalDurationNs", ascending=False)
tot = df.TotalDurationNs.sum()
rf = b["roofline"]
timed = b["steps"] + b["warmup"]
with open(stem + ".md", "w") as fp:
    fp.write(f"# rocprofv3 --kernel-trace --stats of bench.py ({tag}, {cfgname}, {dtype})\n\n"
             f"Command (MI355X, 1 GPU): `{cmd}`\n\n"
             f"Workload: {b['config']['workload']}.  bench.py under the profiler: {b['ms_per_step']:.3f} ms/step "
             f"({b['value'] / 1e6:.1f} M layer-edges/s, launch = {b['config'].get('launch')}); step_ms {json.dumps(b.get('step_ms'))}.\n\n"
             f"The table holds EVERY launch of the run: {b['warmup']} warm-up steps (eager; the last two fully instrumented), capture "
             f"warm-ups, {b['steps']} timed HIP-graph replays and the eagerly issued steps of the roofline leg.  Kernel time summed over "
             f"the run: {tot / 1e6:.1f} ms in {int(df.Calls.sum())} launches.\n\n"
             f"`roofline` of the same run: {rf['kernel']}: bound {rf['bound']}, achieved {rf['achieved']:.1f} {rf['unit']} of {rf['peak']:.0f} "
             f"(frac {rf['frac']:.3f}" + (f", executed MFMA frac {rf['executed_mfma_frac']:.3f}" if 'executed_mfma_frac' in rf else "")
             + f"), {rf['launches_per_step']:.0f} launches/step, average launch {rf['avg_launch_ms'] * 1e3:.1f} us by HIP events on the launch "
             f"stream; PMC traffic per step {('%.3f GB' % (rf['traffic'] / 1e9)) if rf.get('traffic') else 'n/a'}.  Compare `avg us` of the "
             f"same kernels below (shapes are mixed per kernel name; the per-shape averages are in the bench JSON next to this file).\n\n"
             "| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for _, r in df.head(60).iterrows():
        nm = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:110]
        fp.write(f"| `{nm}` | {r['Calls']} | {r['TotalDurationNs'] / 1e6:.3f} | {r['AverageNs'] / 1e3:.1f} | {100 * r['TotalDurationNs'] / tot:.2f} |\n")
print(open(stem + ".md").read()[:1800])
