#!/bin/bash
# Round-4 GPU call 7: does ONE arena (all arrays of a run carved out of one device allocation) remove the process-to-process spread of the write-heavy kernels?
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_7
mkdir -p $OUT
cd $REPO
run() {  # run <workload> <arena GB> <steps> <repeats>
  OTMB_BENCH_ARENA_GB=$2 timeout -k 10 300 python3 bench.py --workload $1 --extra-configs= --no-cpu-baseline --no-end-to-end --warmup 3 --steps $3 --repeats $4 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'workload':'$1','arena_gb':$2,'ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
}
for r in 1 2 3 4 5; do
  run quarterdeg 0 10 2 | tee -a $OUT/arena.jsonl
  run quarterdeg 90 10 2 | tee -a $OUT/arena.jsonl
done
for r in 1 2 3 4; do
  run access1deg 0 20 5 | tee -a $OUT/arena.jsonl
  run access1deg 8 20 5 | tee -a $OUT/arena.jsonl
done
echo "== done =="
