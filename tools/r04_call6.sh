#!/bin/bash
# Round-4 GPU call 6: does a longer warm-up change the measured step?  (call 5: a default bench run measured the 1 degree fill pass at 0.355 ms and the
# 0.25 degree child at 7.29 ms; the profile runs on the same box minutes later 0.3225 / 6.02 ms.)  Fresh processes, alternating warm-up lengths.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_6
mkdir -p $OUT
cd $REPO
run() {  # run <workload> <warmup> <steps> <repeats>
  timeout -k 10 300 python3 bench.py --workload $1 --extra-configs= --no-cpu-baseline --no-end-to-end --warmup $2 --steps $3 --repeats $4 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'workload':'$1','warmup':$2,'steps':$3,'repeats':$4,'ms_per_step':round(d['ms_per_step'],4),'min':round(d['repeats']['ms_per_step_min'],4),'max':round(d['repeats']['ms_per_step_max'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
}
for r in 1 2 3; do
  run access1deg 3 20 5 | tee -a $OUT/warmup.jsonl
  run access1deg 2000 20 5 | tee -a $OUT/warmup.jsonl
  run access1deg 3 2000 3 | tee -a $OUT/warmup.jsonl
done
for r in 1 2; do
  run quarterdeg 2 10 2 | tee -a $OUT/warmup.jsonl
  run quarterdeg 100 10 2 | tee -a $OUT/warmup.jsonl
done
echo "== done =="
