#!/bin/bash
# Round-4 GPU call 17: four-row facefluxes workgroups with the south row through LDS: parity (forced on every grid), A/B at 0.25 degree, L1 counters.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_17
mkdir -p $OUT
cd $REPO
stop() { echo "STOP: $1 (rc=$2)"; exit 1; }
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop "$1" $rc; fi; }
OTMB_FF_ROWS=4 timeout -k 10 800 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_rows4_lds.log 2>&1; rc=$?
tail -4 $OUT/pytest_gpu_rows4_lds.log
if [ $rc -ne 0 ]; then stop "gpu tests" $rc; fi
BARGS="--extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3"
fresh() {  # fresh <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 240 python3 bench.py --workload $wl $BARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
}
for r in 1 2 3; do
  fresh quarterdeg rows4_global OTMB_FF_LDS_SOUTH=0 | tee -a $OUT/fresh_ff_lds.jsonl; guard fresh
  fresh quarterdeg rows4_lds OTMB_FF_LDS_SOUTH=1 | tee -a $OUT/fresh_ff_lds.jsonl; guard fresh
done
cd /tmp && export TMPDIR=/tmp
for lds in 0 1; do
  export OTMB_FF_LDS_SOUTH=$lds
  i=0
  for set in FETCH_SIZE "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/ff_lds$lds/pmc_$i -- python3 $REPO/bench.py --workload quarterdeg --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs= > /dev/null 2> $OUT/ff_lds${lds}_$i.err; guard "pmc lds $lds $set"
  done
  python3 $REPO/tools/pmc_summary.py $OUT/ff_lds$lds "facefluxes" > $OUT/ff_lds${lds}_summary.txt
  rm -rf $OUT/ff_lds$lds
  echo "--- lds=$lds"; cat $OUT/ff_lds${lds}_summary.txt
done
echo "== done =="
