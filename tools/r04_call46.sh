#!/bin/bash
# Round-4 GPU call 46: do the request rates scale with the number of busy CUs (a CU's limit) or saturate (a shared one)?  16 ... 2048 persistent workgroups.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_46
mkdir -p $OUT
cd /tmp
hipcc --offload-arch=gfx950 -O3 -o request_rate $REPO/tools/micro/request_rate.hip 2> $OUT/build.err || { echo "STOP build"; exit 1; }
timeout -k 5 120 ./request_rate 200 > $OUT/cu_scaling.log 2>&1; rc=$?
cat $OUT/cu_scaling.log
echo "rc=$rc"
