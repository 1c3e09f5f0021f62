#!/bin/bash
# Round-4 GPU call 26: the whole GPU suite with the column-block tile order as default and the new bench.py tests (plain; one rank over RCCL).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_26
mkdir -p $OUT
cd $REPO
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -15 $OUT/pytest_gpu.log
echo "rc=$rc"
