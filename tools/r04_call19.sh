#!/bin/bash
# Round-4 GPU call 19: facefluxes four-row workgroups, west neighbour by DPP on top of the south row through LDS (OTMB_FF_LDS_SOUTH=2): parity, A/B at 0.25 degree.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_19
mkdir -p $OUT
cd $REPO
OTMB_FF_ROWS=4 OTMB_FF_LDS_SOUTH=2 timeout -k 10 800 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -3 $OUT/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "STOP tests rc=$rc"; exit 1; fi
BARGS="--extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3"
fresh() {  # fresh <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 240 python3 bench.py --workload $wl $BARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
}
for r in 1 2 3; do
  fresh quarterdeg lds_south OTMB_FF_LDS_SOUTH=1 | tee -a $OUT/fresh_ff_dpp.jsonl
  fresh quarterdeg lds_south_dpp_west OTMB_FF_LDS_SOUTH=2 | tee -a $OUT/fresh_ff_dpp.jsonl
done
echo "== done =="
