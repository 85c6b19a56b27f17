"""Copy a rocprofv3 kernel_stats.csv into profiles/ with a markdown summary. usage: save_profile.py <dir> <steps> <tag> <bench_json>"""
import glob, json, re, shutil, sys
import pandas as pd
d, steps, tag, bj = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
f = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(f, f"profiles/{tag}_kernel_stats_st_pgat_spgnn_3_b512.csv")
shutil.copy(bj, f"profiles/{tag}_bench_under_rocprof.json")
b = json.load(open(bj))
df = pd.read_csv(f).sort_values("TotalDurationNs", ascending=False)
tot = df.TotalDurationNs.sum()
with open(f"profiles/{tag}_kernel_stats_st_pgat_spgnn_3_b512.md", "w") as fp:
    fp.write(f"# rocprofv3 --kernel-trace --stats, round 1, state at the end of the round\n\n"
             f"Command (MI355X, 1 GPU): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --no-cpu-baseline`\n\n"
             f"{steps} optimizer steps ({b['steps']} timed + {b['warmup']} warm-up, 2 of them instrumented) of st_pgat_spgnn_3, 512 trees (N=76410, E=228206), fp32 parity path "
             f"(split-fp16 MFMA GEMMs), dropout on.  bench.py under the profiler: {b['ms_per_step']:.2f} ms/step "
             f"({b['value']/1e6:.1f} M layer-edges/s); kernel time summed: {tot/1e6/steps:.2f} ms/step, {df.Calls.sum()/steps:.0f} launches/step.\n"
             f"Dominant hand-written HBM-bound kernel in bench.py's `roofline`: `{b['roofline']['kernel']}` "
             f"{b['roofline']['avg_launch_ms']*1e3:.1f} us per launch by HIP events (see the same kernel's average below).\n\n"
             "| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for _, r in df.head(45).iterrows():
        nm = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:110]
        fp.write(f"| `{nm}` | {r['Calls']} | {r['TotalDurationNs']/1e6:.3f} | {r['AverageNs']/1e3:.1f} | {100*r['TotalDurationNs']/tot:.2f} |\n")
print(open(f"profiles/{tag}_kernel_stats_st_pgat_spgnn_3_b512.md").read()[:1500])
