#!/bin/bash
# Round-4 GPU call 20: the headline's distribution over fresh processes on one box (final sources): ten runs of the headline child as bench.py starts it.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_20
mkdir -p $OUT
cd $REPO
for r in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 240 python3 bench.py --extra-configs "" --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'run':$r,'value':round(d['value']/1e9,4),'ms_per_step':round(d['ms_per_step'],4),'fill':round(d['kernels_ms']['tm_kernel<fill>'],4),'ff':round(d['kernels_ms']['facefluxes_kernel'],4),'count':round(d['kernels_ms']['tm_count_kernel'],4),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUT/headline_x10.jsonl
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
done
for r in 1 2 3 4 5; do
  timeout -k 10 240 python3 bench.py --workload quarterdeg --extra-configs "" --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 --repeats 2 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'run':$r,'workload':'quarterdeg','value':round(d['value']/1e9,4),'ms_per_step':round(d['ms_per_step'],4),'fill':round(d['kernels_ms']['tm_kernel<fill>'],4),'ff':round(d['kernels_ms']['facefluxes_kernel'],4),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUT/quarterdeg_x5.jsonl
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
done
echo "== done =="
