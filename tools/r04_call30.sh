#!/bin/bash
# Round-4 GPU call 30: the fill pass's time against the allocation its output arrays live in, inside one process (tools/placement_outputs.py), two processes per size.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_30
mkdir -p $OUT
cd $REPO
for r in 1 2; do
  timeout -k 10 200 python3 tools/placement_outputs.py access1deg 8 3 2> $OUT/err_1deg_$r.log | tee -a $OUT/placement_outputs.jsonl || { echo STOP; tail -5 $OUT/err_1deg_$r.log; exit 1; }
done
for r in 1 2; do
  timeout -k 10 300 python3 tools/placement_outputs.py quarterdeg 5 2 2> $OUT/err_qdeg_$r.log | tee -a $OUT/placement_outputs.jsonl || { echo STOP; tail -5 $OUT/err_qdeg_$r.log; exit 1; }
done
