#!/bin/bash
# The headline workload dense / loss_rows_only="backward" / loss_rows_only=True on ONE box: bench lines, and the kernel statistics
# of the last under rocprofv3.  usage: bash tools/loss_rows_measure.sh <tag>
TAG=${1:-r05d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for m in dense backward all; do
  case $m in dense) F="";; backward) F="--loss-rows-backward";; all) F="--loss-rows-only";; esac
  python bench.py --full-line --no-cpu-baseline --no-secondary $F > $O/$m.log 2> $O/$m.err || echo "bench $m failed"
  grep '^{' $O/$m.log | tail -1 > $O/${TAG}_bench_loss_rows_$m.json
done
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_all -- python3 bench.py --full-line --no-cpu-baseline --no-secondary --loss-rows-only > $O/prof_all.log 2> $O/prof_all.err) || echo "rocprof failed"
grep '^{' $O/prof_all.log | tail -1 > $O/prof_all.json
(cd $R && python3 tools/save_profile.py $O/prof_all ${TAG}_loss_rows $O/prof_all.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary --loss-rows-only" > $O/prof_all.summary.txt) || true
(cd $R && python3 tools/trace_steps.py $O/prof_all > $O/${TAG}_step_sequence_loss_rows_all.txt 2> /dev/null) || true
cd $R
cp profiles/${TAG}_loss_rows* $O/ 2>/dev/null
python - <<P
import json, glob, os
for f in sorted(glob.glob("$O/${TAG}_bench_loss_rows_*.json")):
    d = json.load(open(f)); print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"] / 1e6, 1), d["loss"])
P
