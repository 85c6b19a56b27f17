#!/bin/bash
# bench.py (f32 headline, bf16 st_gat_6) with each library under build/variants/ named on the command line copied over the
# product .so, back to back on one box ("orig" = the product library itself).  usage: bench_variants.sh orig <name> ...
cd $GRAFT_REPO_ROOT
cp spgnn_amd/libspgnn_hip.so /tmp/orig.so
for v in "$@"; do
  if [ $v != orig ]; then cp build/variants/$v.so spgnn_amd/libspgnn_hip.so; else cp /tmp/orig.so spgnn_amd/libspgnn_hip.so; fi
  python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bv_$v.json
  python bench.py --no-cpu-baseline --config st_gat_6 --dtype bf16 2>/dev/null | grep '^{' | tail -1 > gpurun_out/bv_${v}_bf16.json
done
cp /tmp/orig.so spgnn_amd/libspgnn_hip.so
python - "$@" <<'P'
import json, sys
for v in sys.argv[1:]:
    for suf in ("", "_bf16"):
        d = json.load(open(f"gpurun_out/bv_{v}{suf}.json")); k = d["roofline_k123"]; g = d["gemm"]
        print(v + suf, round(d["ms_per_step"], 3), "k123 ms", round(k["ms_per_step"], 4), "mp ms", round(d["message_passing"]["ms_per_step"], 4),
              "gemm ms", round(g["ms_per_step"], 3))
P
