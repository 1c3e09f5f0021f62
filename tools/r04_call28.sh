#!/bin/bash
# Round-4 GPU call 28: what does the HBM write path like?  8 GiB written per launch: plain / non-temporal, stores in flight, streams, grid size.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_28
mkdir -p $OUT
cd /tmp
hipcc --offload-arch=gfx950 -O3 -o request_rate $REPO/tools/micro/request_rate.hip 2> $OUT/build.err || { echo "STOP build"; exit 1; }
timeout -k 5 120 ./request_rate 100 > $OUT/write_sweep.log 2>&1; rc=$?
cat $OUT/write_sweep.log
echo "rc=$rc"
