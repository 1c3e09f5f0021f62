#!/bin/bash
# Round-4 GPU call 27: L1 -> L2 request rates (L2 hits / HBM reads / writes).  (The CU-mask legs of the first version hang on this pool: see tools/micro/request_rate.hip.)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_27
mkdir -p $OUT
cd /tmp
hipcc --offload-arch=gfx950 -O3 -o request_rate $REPO/tools/micro/request_rate.hip 2> $OUT/build.err || { echo "STOP build"; exit 1; }
timeout -k 5 60 ./request_rate 0 > $OUT/request_rate.log 2>&1; rc=$?
cat $OUT/request_rate.log
echo "rc=$rc"
