#!/bin/bash
# usage: ab_env.sh VAR v1 v2 ... ; interleaved bench runs, two rounds
VAR=$1; shift
for i in 1 2; do for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],3))"
done; done
