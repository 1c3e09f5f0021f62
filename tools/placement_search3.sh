#!/bin/bash
# Placement search 3 (round 4): physical granularity.  torch's expandable segments map every tensor's range from separate 20 MB physical handles (hipMemCreate / hipMemMap).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUTF=$REPO/gpurun_out/placement_search3.jsonl
mkdir -p $(dirname $OUTF)
cd $REPO
run() {  # run <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload $wl --extra-configs= --no-cpu-baseline --no-end-to-end --warmup 3 --steps 10 --repeats 3 2> $REPO/gpurun_out/ps3_last.err | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'workload':'$wl','tag':'$tag','ms_per_step':round(d['ms_per_step'],4),'fill':round(d['kernels_ms']['tm_kernel<fill>'],4),'ff':round(d['kernels_ms']['facefluxes_kernel'],4),'count':round(d['kernels_ms']['tm_count_kernel'],4),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUTF
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
}
E="expandable_segments:True"
for r in 1 2 3; do
  run access1deg separate X=0
  run access1deg expandable PYTORCH_ALLOC_CONF=$E PYTORCH_CUDA_ALLOC_CONF=$E PYTORCH_HIP_ALLOC_CONF=$E
done
tail -3 $REPO/gpurun_out/ps3_last.err
for r in 1 2; do
  run quarterdeg separate X=0
  run quarterdeg expandable PYTORCH_ALLOC_CONF=$E PYTORCH_CUDA_ALLOC_CONF=$E PYTORCH_HIP_ALLOC_CONF=$E
done
echo "== done =="
