#!/bin/bash
# Round-4 GPU call 37: the supporting kernels on the 1 degree grid: time, algorithmic bytes, fraction of the HBM peak (tools/secondary_time.py).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_37
mkdir -p $OUT
cd $REPO
timeout -k 10 400 python3 tools/secondary_time.py access1deg > $OUT/secondary_access1deg.jsonl 2> $OUT/err.log; rc=$?
cat $OUT/secondary_access1deg.jsonl | cut -c1-260 | head -30
echo "rc=$rc"; tail -5 $OUT/err.log
