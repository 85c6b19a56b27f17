#!/bin/bash
# HBM bytes per launch of the message-passing kernels: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over
# tools/mp_driver.py, merged into profiles/traffic_latest.json (keys = bench.py's kernel keys).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mp; mkdir -p $O
(cd $R && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/mp_driver.py > $O/fetch.log 2>&1)
(cd $R && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 tools/mp_driver.py > $O/write.log 2>&1)
cd $R && python3 tools/pmc_traffic.py $O/fetch $O/write gpurun_out/mp_manifest.json $O/mp_traffic.json $O/mp_traffic.md | tail -5
