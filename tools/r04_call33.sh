#!/bin/bash
# Round-4 GPU call 33: DeviceAssembler.choose_placement: parity test, then fresh-process A/B of bench.py with 1 (off) and 4 candidates at 0.25 and 1 degree.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_33
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_formulations.py tests/test_bench_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then echo "STOP tests rc=$rc"; tail -30 $OUT/pytest.log; exit 1; fi
fresh() {  # fresh <workload> <candidates>
  timeout -k 10 400 python3 bench.py --workload $1 --extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3 --placement-candidates $2 2> $OUT/err.log | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'workload':'$1','candidates':$2,'ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4),'placement':d.get('placement')}))"
  rc=$?; if [ $rc -ne 0 ]; then echo "STOP rc=$rc"; tail -5 $OUT/err.log; exit 1; fi
}
for r in 1 2 3; do
  fresh quarterdeg 1 | tee -a $OUT/fresh_placement.jsonl
  fresh quarterdeg 4 | tee -a $OUT/fresh_placement.jsonl
done
for r in 1 2 3; do
  fresh access1deg 1 | tee -a $OUT/fresh_placement.jsonl
  fresh access1deg 4 | tee -a $OUT/fresh_placement.jsonl
done
fresh access1deg 8 | tee -a $OUT/fresh_placement.jsonl
echo "== done =="
