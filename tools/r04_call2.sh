#!/bin/bash
# Round-4 GPU call 2: parity (new: otmb_mgpu_*, host pool, bolus at 1 degree, heavy tiles dealt over the XCDs), A/B of the dealing,
# timeline with it, HBM traffic of facefluxes / count under both mappings (one counter per pass: two together exceed the hardware).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_2
mkdir -p $OUT
cd $REPO
stop() { echo "STOP: $1 (rc=$2)"; exit 1; }
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop "$1" $rc; fi; }

echo "== gpu tests =="
timeout -k 10 800 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -8 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop "gpu tests" $rc; fi   # (a failing test is reported; the measurements below still run)

BARGS="--extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3"
fresh() {  # fresh <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 240 python3 bench.py --workload $wl $BARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
}
echo "== fresh-process A/B: heavy tiles dealt over the XCDs =="
for r in 1 2 3; do
  fresh access1deg all_in_xcd0 OTMB_DEAL_HEAVY=0 | tee -a $OUT/fresh_access1deg.jsonl; guard fresh
  fresh access1deg dealt OTMB_DEAL_HEAVY=1 | tee -a $OUT/fresh_access1deg.jsonl; guard fresh
done
for r in 1 2; do
  fresh quarterdeg all_in_xcd0 OTMB_DEAL_HEAVY=0 | tee -a $OUT/fresh_quarterdeg.jsonl; guard fresh
  fresh quarterdeg dealt OTMB_DEAL_HEAVY=1 | tee -a $OUT/fresh_quarterdeg.jsonl; guard fresh
done

echo "== dispatch timeline of the fill pass =="
OTMB_STAMPS_PREBUILT=1 OTMB_DEAL_HEAVY=0 timeout -k 10 200 python3 tools/stamps.py access1deg > $OUT/timeline_access1deg_all_in_xcd0.log 2>&1; guard stamps
OTMB_STAMPS_PREBUILT=1 OTMB_DEAL_HEAVY=1 timeout -k 10 200 python3 tools/stamps.py access1deg > $OUT/timeline_access1deg_dealt.log 2>&1; guard stamps
cat $OUT/timeline_access1deg_all_in_xcd0.log; cat $OUT/timeline_access1deg_dealt.log

cd /tmp && export TMPDIR=/tmp
echo "== HBM traffic of facefluxes / count, old and new mapping =="
for wl in access1deg quarterdeg; do
  for tag in old new cnt2; do
    if [ $tag = old ]; then export OTMB_FF_XCD=0 OTMB_COUNT_ORDER=0; elif [ $tag = new ]; then export OTMB_FF_XCD=1 OTMB_COUNT_ORDER=1; else export OTMB_FF_XCD=1 OTMB_COUNT_ORDER=2; fi
    i=0
    for set in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/traffic_${wl}_$tag/pmc_$i -- python3 $REPO/bench.py --workload $wl --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs= > /dev/null 2> $OUT/traffic_${wl}_${tag}_$i.err; guard "pmc $wl $tag $set"
    done
    python3 $REPO/tools/pmc_summary.py $OUT/traffic_${wl}_$tag "tm_kernel,tm_count,facefluxes" > $OUT/traffic_${wl}_${tag}_summary.txt
    rm -rf $OUT/traffic_${wl}_$tag
    echo "--- $wl $tag"; cat $OUT/traffic_${wl}_${tag}_summary.txt
  done
done
unset OTMB_FF_XCD OTMB_COUNT_ORDER
echo "== done =="
