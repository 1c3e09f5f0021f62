#!/bin/bash
# Round-4 GPU call 21: (1) the fill pass's run stores on the 1 KB grid of the destination (-DOTMB_ALIGN1K, variant library): parity + fresh A/B;
# (2) the traffic model's "line-aligned chunks" lines; (3) per-channel spread of the L2 requests of the fill pass (raw TCC counters, JSON output).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_21
mkdir -p $OUT
cd $REPO
V=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_align1k.so
OTMB_LIB_OVERRIDE=$V timeout -k 10 800 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu_align1k.log 2>&1; rc=$?
tail -5 $OUT/pytest_gpu_align1k.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "STOP tests timed out"; exit 1; fi
echo "== fresh A/B 1 degree =="
timeout -k 10 500 bash tools/fresh_ab.sh access1deg 3 default align1k | tee $OUT/fresh_ab_access1deg.txt || exit 1
echo "== fresh A/B 0.25 degree =="
timeout -k 10 500 bash tools/fresh_ab.sh quarterdeg 2 default align1k | tee $OUT/fresh_ab_quarterdeg.txt || exit 1
echo "== traffic model =="
( cd /tmp && hipcc --offload-arch=gfx950 -O3 -o fill_model $REPO/tools/micro/fill_model.hip 2> $OUT/fill_model_build.err && timeout -k 10 200 ./fill_model > $OUT/fill_model_access1deg.log 2>&1 ) || { echo "STOP model"; exit 1; }
grep -n "shifted\|aligned\|plain: loads" $OUT/fill_model_access1deg.log
echo "== per-channel L2 requests =="
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs="
timeout -k 10 200 rocprofv3 --pmc TCC_REQ TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace --output-format json -d $OUT/tcc_1deg -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/tcc_1deg.err
echo "tcc pass rc=$?"
python3 $REPO/tools/tcc_channels.py $OUT/tcc_1deg tm_kernel > $OUT/tcc_channels_access1deg.json 2> $OUT/tcc_channels.err
ls -la $OUT/tcc_1deg | head; du -sh $OUT/tcc_1deg
python3 - <<EOF
import json
d=json.load(open("$OUT/tcc_channels_access1deg.json"))
print(d.get("example_record_keys"), d.get("counter_meta_sample"))
for k,v in d["kernels"].items():
    for c,x in v.items():
        print(k[:40], c, x["instances"], x["dispatches"], round(x["sum"]), x["min"], x["max"], x["max_over_mean"])
EOF
# keep one raw JSON if small enough, otherwise drop the raw directory
find $OUT/tcc_1deg -name "*.json" -size +20M -delete
echo "== done =="
