#!/bin/bash
# Round-4 GPU call 18: profiles of the FINAL kernels (traffic.json is keyed to the kernel sources) + four-row / LDS facefluxes forced at 1 degree.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_18
mkdir -p $OUT
cd $REPO
BARGS="--extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3"
fresh() {  # fresh <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 240 python3 bench.py --workload $wl $BARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
}
for r in 1 2 3; do
  fresh access1deg rows1 OTMB_FF_ROWS=1 | tee -a $OUT/fresh_ff_1deg.jsonl
  fresh access1deg rows4_lds OTMB_FF_ROWS=4 | tee -a $OUT/fresh_ff_1deg.jsonl
done
echo "== profile 1 degree =="
timeout -k 10 900 bash tools/profile.sh r04b_1deg > $OUT/profile_1deg.log 2>&1; rc=$?; tail -2 $OUT/profile_1deg.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
echo "== profile 0.25 degree =="
timeout -k 10 1100 bash tools/profile.sh r04b_qdeg --workload quarterdeg > $OUT/profile_qdeg.log 2>&1; rc=$?; tail -2 $OUT/profile_qdeg.log
echo "== done =="
