#!/bin/bash
# Round-4 GPU call 34: the 1 degree passes against the allocations of their INPUT arrays, inside one process (tools/placement_inputs.py), three processes.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_34
mkdir -p $OUT
cd $REPO
for r in 1 2 3; do
  timeout -k 10 200 python3 tools/placement_inputs.py 6 2 2> $OUT/err_$r.log | tee -a $OUT/placement_inputs.jsonl | python3 -c "
import json,sys
d=json.load(sys.stdin)['fill_ff_count_ms_by_input_copy']
print({k:[x['tm_kernel<fill>'] for x in v] for k,v in d.items()})
print({k:[x['facefluxes_kernel'] for x in v] for k,v in d.items()})" || { echo STOP; tail -5 $OUT/err_$r.log; exit 1; }
done
