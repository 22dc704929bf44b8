#!/bin/bash
# HBM traffic of the product kernels from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes: FETCH_SIZE
# and WRITE_SIZE do not fit one pass; no trace domains besides --kernel-trace).  Run on the GPU box from the repo root:
#   bash tools/collect_traffic.sh    -> gpurun_out/traffic/{fetch,write}/..., then tools/traffic_summary.py
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAMES="dec1.dg dec0.dg enc5.fwd dec0.fwd0 dec0.fwd0.wg enc4.fwd.wg dec1.fwd0.wg"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic/fetch -- python3 $R/tools/prof_one.py $NAMES --reps 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic/write -- python3 $R/tools/prof_one.py $NAMES --reps 3 > /dev/null 2>&1
echo collected
