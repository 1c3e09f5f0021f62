#!/bin/bash
# Round-4 GPU call 49: facefluxes with plain instead of non-temporal stores on the large grids (OTMB_FF_NT), now that the south row comes through LDS: fresh-process A/B of the whole step.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_49
mkdir -p $OUT
cd $REPO
fresh() {  # fresh <workload> <tag> <steps> ENV...
  wl=$1; tag=$2; st=$3; shift; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload $wl --extra-configs= --no-cpu-baseline --no-end-to-end --steps $st --warmup 2 --repeats 2 --placement-candidates 1 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()}}))"
  rc=$?; if [ $rc -ne 0 ]; then echo "($wl $tag rc=$rc)"; fi
}
for r in 1 2 3; do
  fresh quarterdeg nt 10 OTMB_FF_NT=1 | tee -a $OUT/fresh_ff_nt.jsonl
  fresh quarterdeg plain 10 OTMB_FF_NT=0 | tee -a $OUT/fresh_ff_nt.jsonl
done
for r in 1 2; do
  fresh tenthdeg nt 4 OTMB_FF_NT=1 | tee -a $OUT/fresh_ff_nt.jsonl
  fresh tenthdeg plain 4 OTMB_FF_NT=0 | tee -a $OUT/fresh_ff_nt.jsonl
done
for r in 1 2; do
  fresh access1deg plain_default 10 OTMB_FF_NT=0 | tee -a $OUT/fresh_ff_nt.jsonl
  fresh access1deg nt 10 OTMB_FF_NT=1 | tee -a $OUT/fresh_ff_nt.jsonl
done
