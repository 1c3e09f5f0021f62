#!/bin/bash
# Round-4 GPU call 1: parity with the new defaults (facefluxes / counting pass in XCD-contiguous eighths, index prefetch in the fill pass),
# in-process and fresh-process A/B of each knob, dispatch timeline of the fill pass, HBM traffic of facefluxes / count.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_1
mkdir -p $OUT
cd $REPO
stop() { echo "STOP: $1 (rc=$2)"; exit 1; }
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop "$1" $rc; fi; }

echo "== gpu tests =="
timeout -k 10 560 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -3 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || stop "gpu tests" $rc

echo "== in-process A/B at 1 degree =="
timeout -k 10 300 python3 tools/env_ab.py --workload access1deg --reps 2 --rounds 3 --steps 20 --variants \
 "base:OTMB_FF_XCD=0,OTMB_COUNT_ORDER=0,OTMB_PF_DIST=0;ffxcd:OTMB_FF_XCD=1,OTMB_COUNT_ORDER=0,OTMB_PF_DIST=0;cnt1:OTMB_FF_XCD=0,OTMB_COUNT_ORDER=1,OTMB_PF_DIST=0;cnt2:OTMB_FF_XCD=0,OTMB_COUNT_ORDER=2,OTMB_PF_DIST=0;pf48:OTMB_FF_XCD=0,OTMB_COUNT_ORDER=0,OTMB_PF_DIST=48;pf96:OTMB_FF_XCD=0,OTMB_COUNT_ORDER=0,OTMB_PF_DIST=96;pf192:OTMB_FF_XCD=0,OTMB_COUNT_ORDER=0,OTMB_PF_DIST=192;all:OTMB_FF_XCD=1,OTMB_COUNT_ORDER=1,OTMB_PF_DIST=96" \
 > $OUT/env_ab_access1deg.jsonl 2> $OUT/env_ab_access1deg.err; guard "env_ab access1deg"
cat $OUT/env_ab_access1deg.jsonl

BARGS="--extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3"
fresh() {  # fresh <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 240 python3 bench.py --workload $wl $BARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
}
echo "== fresh-process A/B at 1 degree =="
for r in 1 2 3; do
  fresh access1deg base OTMB_FF_XCD=0 OTMB_COUNT_ORDER=0 OTMB_PF_DIST=0 | tee -a $OUT/fresh_access1deg.jsonl; guard fresh
  fresh access1deg new OTMB_X=0 | tee -a $OUT/fresh_access1deg.jsonl; guard fresh
  fresh access1deg new_pf0 OTMB_PF_DIST=0 | tee -a $OUT/fresh_access1deg.jsonl; guard fresh
done
echo "== fresh-process A/B at 0.25 degree =="
for r in 1 2; do
  fresh quarterdeg base OTMB_FF_XCD=0 OTMB_COUNT_ORDER=0 OTMB_PF_DIST=0 | tee -a $OUT/fresh_quarterdeg.jsonl; guard fresh
  fresh quarterdeg new OTMB_X=0 | tee -a $OUT/fresh_quarterdeg.jsonl; guard fresh
  fresh quarterdeg new_pf0 OTMB_PF_DIST=0 | tee -a $OUT/fresh_quarterdeg.jsonl; guard fresh
  fresh quarterdeg new_cnt2 OTMB_COUNT_ORDER=2 | tee -a $OUT/fresh_quarterdeg.jsonl; guard fresh
done

echo "== dispatch timeline of the fill pass (stamps build) =="
OTMB_STAMPS_PREBUILT=1 OTMB_PF_DIST=0 timeout -k 10 200 python3 tools/stamps.py access1deg > $OUT/timeline_access1deg_pf0.log 2>&1; guard stamps
OTMB_STAMPS_PREBUILT=1 OTMB_PF_DIST=96 timeout -k 10 200 python3 tools/stamps.py access1deg > $OUT/timeline_access1deg_pf96.log 2>&1; guard stamps
cat $OUT/timeline_access1deg_pf0.log; cat $OUT/timeline_access1deg_pf96.log

cd /tmp && export TMPDIR=/tmp
echo "== HBM traffic of facefluxes / count, old and new mapping =="
for wl in access1deg quarterdeg; do
  for tag in old new; do
    if [ $tag = old ]; then E="OTMB_FF_XCD=0 OTMB_COUNT_ORDER=0"; else E="OTMB_FF_XCD=1 OTMB_COUNT_ORDER=1"; fi
    i=0
    for set in "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      export $E
      timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/traffic_${wl}_$tag/pmc_$i -- python3 $REPO/bench.py --workload $wl --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs= > /dev/null 2> $OUT/traffic_${wl}_${tag}_$i.err; guard "pmc $wl $tag"
    done
    python3 $REPO/tools/pmc_summary.py $OUT/traffic_${wl}_$tag "tm_kernel,tm_count,facefluxes" > $OUT/traffic_${wl}_${tag}_summary.txt
    rm -rf $OUT/traffic_${wl}_$tag
    echo "--- $wl $tag"; cat $OUT/traffic_${wl}_${tag}_summary.txt
  done
done
echo "== done =="
