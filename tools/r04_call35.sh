#!/bin/bash
# Round-4 GPU call 35: knobs tuned at 1 / 0.25 degree only, on the 0.1 degree grid (counting-pass order, facefluxes rows / south row through LDS); two tiles per counting workgroup at all sizes.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_35
mkdir -p $OUT
cd $REPO
fresh() {  # fresh <workload> <tag> <steps> ENV...
  wl=$1; tag=$2; st=$3; shift; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload $wl --extra-configs= --no-cpu-baseline --no-end-to-end --steps $st --warmup 2 --repeats 2 --placement-candidates 1 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()}}))"
  rc=$?; if [ $rc -ne 0 ]; then echo "STOP $wl $tag rc=$rc"; exit 1; fi
}
V=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_tpb2.so
for r in 1 2; do
  fresh tenthdeg default 4 OTMB_X=0 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg count_order0 4 OTMB_COUNT_ORDER=0 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg count_order1 4 OTMB_COUNT_ORDER=1 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg ff_rows1 4 OTMB_FF_ROWS=1 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg ff_lds_south0 4 OTMB_FF_LDS_SOUTH=0 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg count_tpb2 4 OTMB_LIB_OVERRIDE=$V | tee -a $OUT/fresh_tenthdeg.jsonl
done
for r in 1 2; do
  fresh quarterdeg default 10 OTMB_X=0 | tee -a $OUT/fresh_other.jsonl
  fresh quarterdeg count_tpb2 10 OTMB_LIB_OVERRIDE=$V | tee -a $OUT/fresh_other.jsonl
  fresh access1deg default 10 OTMB_X=0 | tee -a $OUT/fresh_other.jsonl
  fresh access1deg count_tpb2 10 OTMB_LIB_OVERRIDE=$V | tee -a $OUT/fresh_other.jsonl
done
echo "== done =="
