#!/bin/bash
# kernel trace of the bench: usage tools/r03_prof.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
T=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/prof_$T; mkdir -p $O
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 6 "$@" > $O.log 2> $O.err
grep '^{' $O.log | tail -1 > $O.json
python3 - $O <<'P'
import glob, os, sys, re
import pandas as pd
d = sys.argv[1]
f = max(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
df = pd.read_csv(f).sort_values("TotalDurationNs", ascending=False)
with open(d + "_stats.txt", "w") as out:
    for _, r in df.iterrows():
        nm = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Name"])[:110]
        out.write(f"{r['Calls']:6d} {r['TotalDurationNs']/1e6:10.3f} ms {r['AverageNs']/1e3:9.1f} us  {nm}\n")
print(open(d + "_stats.txt").read()[:200])
P
