#!/bin/bash
# Round measurement on one GPU box.  usage: bash tools/round_measure.sh <tag>   (e.g. r04)
#  1. PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one eager step: headline f32 and st_gat_6 bf16 at 512 trees,
#     st_pgat_spgnn_3 and st_gat_3 at 64 trees (BASELINE configs 2 / 3) -> profiles/traffic_latest.json (bench.py reads it
#     for `roofline.traffic`) and profiles/<tag>_pmc_traffic_*.md
#  2. bench.py under rocprofv3 --kernel-trace --stats for both 512-tree configs -> profiles/<tag>_kernel_stats_*.{csv,md}
#  3. the plain bench lines (with the CPU baseline, the secondary legs and the batch cycle) -> profiles/<tag>_bench_*.json
# Everything judged is ALSO copied under gpurun_out/<tag>/profiles/ so it travels back.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O/profiles
cd /tmp && export TMPDIR=/tmp
pmc() {  # <config> <dtype> <trees>
  local c=$1 dt=$2 t=$3 D=$O/pmc_${1}_${2}_${3}; mkdir -p $D
  local sfx=""; if [ "$t" != "512" ]; then sfx="_${t}trees"; fi
  (cd $R && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- python3 tools/pmc_step.py $c $dt $t $D/manifest.json > $D.fetch.log 2>&1) || return 1
  (cd $R && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- python3 tools/pmc_step.py $c $dt $t $D/manifest_w.json > $D.write.log 2>&1) || return 1
  (cd $R && python3 tools/pmc_merge.py $D/fetch $D/write $D/manifest.json profiles/${TAG}_pmc_traffic_${c}_${dt}${sfx}.md profiles/traffic_latest.json)
}
prof() {  # <config> <dtype> [extra bench flag] [file suffix]
  local c=$1 dt=$2 X=$3 S=$4 D=$O/prof_${1}_${2}$4
  local CMD="rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof_${c}_${dt}${S} -- python3 bench.py --full-line --config $c --dtype $dt --no-cpu-baseline --no-secondary $X"
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --full-line --config $c --dtype $dt --no-cpu-baseline --no-secondary $X > $D.log 2> $D.err) || return 1
  grep '^{' $D.log | tail -1 > $D.json
  (cd $R && python3 tools/save_profile.py $D $TAG $D.json "$CMD" $S > $D.summary.txt)
  (cd $R && python3 tools/trace_steps.py $D > profiles/${TAG}_step_sequence_${c}_${dt}${S}.txt 2> /dev/null) || true
}
pmc st_pgat_spgnn_3 f32 512 && pmc st_gat_6 bf16 512 && pmc st_pgat_spgnn_3 f32 64 && pmc st_gat_3 f32 64 && prof st_pgat_spgnn_3 f32 && prof st_pgat_spgnn_3 f32 --no-side-stream _no_side_stream && prof st_gat_6 bf16 || echo "MEASURE STEP FAILED"
echo "profiling done" > $O/progress.txt
cd $R
python bench.py --full-line --no-secondary > $O/bench_f32.log 2> $O/bench_f32.err; grep '^{' $O/bench_f32.log | tail -1 > profiles/${TAG}_bench_st_pgat_spgnn_3_f32.json
python bench.py --full-line --config st_gat_6 --dtype bf16 --no-secondary > $O/bench_bf16.log 2> $O/bench_bf16.err; grep '^{' $O/bench_bf16.log | tail -1 > profiles/${TAG}_bench_st_gat_6_bf16.json
echo "bench lines done" >> $O/progress.txt
for c in st_gat_3 st_gat_6 st_gcn_3 st_gin_3 st_sage_3; do
  python bench.py --full-line --config $c --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | grep '^{' | tail -1 > $O/cfg_${c}_f32.json
done
python bench.py --full-line --trees 64 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | tail -1 > $O/cfg_headline_64trees.json
echo "configs done" >> $O/progress.txt
python bench.py --full-line > $O/bench_default.log 2> $O/bench_default.err; grep '^{' $O/bench_default.log | tail -1 > profiles/${TAG}_bench_default_with_secondary_legs.json
python bench.py --full-line --batch-cycle-only 2> /dev/null | grep '^{' | tail -1 > profiles/${TAG}_batch_cycle_64trees.json
cp profiles/${TAG}_* profiles/traffic_latest.json $O/profiles/
python - <<P
import json, glob, os
for f in sorted(glob.glob("$O/profiles/${TAG}_bench_*.json")) + sorted(glob.glob("$O/cfg_*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline") or {}
        print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"] / 1e6, 1), r.get("frac"), r.get("executed_mfma_frac"), (r.get("hbm") or {}).get("frac"), r.get("traffic"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
