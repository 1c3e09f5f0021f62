#!/bin/bash
# Round-4 GPU call 31: eight output sets at 0.25 degree with every array's address, two processes.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_31
mkdir -p $OUT
cd $REPO
for r in 1 2; do
  timeout -k 10 400 python3 tools/placement_outputs.py quarterdeg 8 2 2> $OUT/err_qdeg_$r.log >> $OUT/placement_outputs.jsonl || { echo STOP; tail -5 $OUT/err_qdeg_$r.log; exit 1; }
done
python3 -c "
import json
for l in open('$OUT/placement_outputs.jsonl'):
    d=json.loads(l); print({k:v for k,v in d['fill_ms_by_output_set'].items()})"
