#!/bin/bash
# A/B of the working tree against an older checkout under build/oldtree (git worktree add build/oldtree <commit>; built there),
# alternating, one box: usage old_vs_new.sh [rounds]
cd $GRAFT_REPO_ROOT
R=${1:-2}
for r in $(seq 1 $R); do
  (cd build/oldtree && python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/ovn_old_$r.json)
  python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > gpurun_out/ovn_new_$r.json
done
python - $R <<'P'
import json, sys
for r in range(1, int(sys.argv[1]) + 1):
    for w in ("old", "new"):
        d = json.load(open(f"gpurun_out/ovn_{w}_{r}.json"))
        print(w, r, round(d["ms_per_step"], 3), "gemm", round(d["gemm"]["ms_per_step"], 3), "mp", round(d["message_passing"]["ms_per_step"], 3),
              "k123", round(d["roofline_k123"]["ms_per_step"], 3), "mfma", round(d["roofline"]["executed_mfma_frac"], 4))
P
