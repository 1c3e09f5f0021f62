#!/bin/bash
# Round-4 GPU call 23: the traffic model's 3-D block variant (inputs staged in LDS by coalesced row loads) against the gather model, both grid sizes.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_23
mkdir -p $OUT
cd /tmp
hipcc --offload-arch=gfx950 -O3 -o fill_model $REPO/tools/micro/fill_model.hip 2> $OUT/build.err || { echo "STOP build"; exit 1; }
timeout -k 10 200 ./fill_model 360 300 50 3051515 quick > $OUT/fill_model_access1deg.log 2>&1 || { echo "STOP 1deg"; tail -5 $OUT/fill_model_access1deg.log; exit 1; }
cat $OUT/fill_model_access1deg.log
timeout -k 10 300 ./fill_model 1440 1080 75 63526216 quick > $OUT/fill_model_quarterdeg.log 2>&1 || { echo "STOP qdeg"; tail -5 $OUT/fill_model_quarterdeg.log; exit 1; }
cat $OUT/fill_model_quarterdeg.log
echo "== done =="
