#!/bin/bash
# Round-4 GPU call 32: per-buffer streaming write / read rates of a dozen separate allocations (tools/micro/buffer_lottery.hip).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_32
mkdir -p $OUT
cd /tmp
hipcc --offload-arch=gfx950 -O3 -o buffer_lottery $REPO/tools/micro/buffer_lottery.hip 2> $OUT/build.err || { echo "STOP build"; exit 1; }
timeout -k 5 120 ./buffer_lottery 12 3400 > $OUT/buffer_lottery_3400MiB.log 2>&1; echo "rc=$?"
cat $OUT/buffer_lottery_3400MiB.log
timeout -k 5 120 ./buffer_lottery 12 160 > $OUT/buffer_lottery_160MiB.log 2>&1; echo "rc=$?"
tail -13 $OUT/buffer_lottery_160MiB.log
