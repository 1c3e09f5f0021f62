#!/bin/bash
# Round-4 GPU call 3: what makes a tile expensive (stamps: tile life against its span), timeline at 0.25 degree, counting-pass order A/B.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_3
mkdir -p $OUT
cd $REPO
stop() { echo "STOP: $1 (rc=$2)"; exit 1; }
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop "$1" $rc; fi; }
OTMB_STAMPS_PREBUILT=1 timeout -k 10 200 python3 tools/stamps.py access1deg > $OUT/timeline_access1deg.log 2>&1; guard stamps
cat $OUT/timeline_access1deg.log
OTMB_STAMPS_PREBUILT=1 timeout -k 10 400 python3 tools/stamps.py quarterdeg > $OUT/timeline_quarterdeg.log 2>&1; guard stamps
cat $OUT/timeline_quarterdeg.log
BARGS="--extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3"
fresh() {  # fresh <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 240 python3 bench.py --workload $wl $BARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
}
for r in 1 2 3; do
  fresh access1deg cnt0 OTMB_COUNT_ORDER=0 | tee -a $OUT/fresh_count.jsonl; guard fresh
  fresh access1deg cnt2 OTMB_COUNT_ORDER=2 | tee -a $OUT/fresh_count.jsonl; guard fresh
done
echo "== done =="
