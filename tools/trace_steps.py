"""Per-step wall time, summed kernel time and launch count from a rocprofv3 kernel trace: steps are delimited by the
optimizer kernel (sgd_momentum).  usage: python tools/trace_steps.py <dir with *kernel_trace.csv> [marker substring]"""
import glob, os, sys
import pandas as pd

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_momentum"
f = max(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
ends = df.index[df["Kernel_Name"].str.contains(marker)].tolist()
rows = []
for a, b in zip(ends[:-1], ends[1:]):
    s = df.iloc[a + 1:b + 1]
    wall = (s["End_Timestamp"].max() - df.loc[a, "End_Timestamp"]) / 1e3
    busy = (s["End_Timestamp"] - s["Start_Timestamp"]).sum() / 1e3
    rows.append((len(s), wall, busy))
r = pd.DataFrame(rows, columns=["launches", "wall_us", "busy_us"])
print(r.groupby("launches").agg(n=("wall_us", "size"), wall_med=("wall_us", "median"), wall_min=("wall_us", "min"),
                                busy_med=("busy_us", "median")).to_string())
# the most common launch count = the captured step: its kernels by time
k = r["launches"].mode()[0]
idx = [i for i, x in enumerate(rows) if x[0] == k]
a, b = ends[idx[len(idx) // 2]], ends[idx[len(idx) // 2] + 1]
s = df.iloc[a + 1:b + 1].copy()
s["dur"] = (s["End_Timestamp"] - s["Start_Timestamp"]) / 1e3
s["gap"] = (s["Start_Timestamp"] - s["End_Timestamp"].shift(1)) / 1e3
s["name"] = s["Kernel_Name"].str.replace(r"\(anonymous namespace\)::|void |at::native::", "", regex=True).str.slice(0, 90)
print(f"-- one step of {k} launches: busy {s['dur'].sum():.1f} us, gaps {s['gap'].iloc[1:].sum():.1f} us")
for _, x in s.iterrows():
    print(f"{x['dur']:8.1f} {x['gap']:7.1f}  {x['name']}")
