#!/bin/bash
# Round-4 GPU call 48: two / one workgroup(s) of the fill pass per CU instead of three (LDS padding, -DOTMB_LDS_PAD; call 47 used -DTM_WAVES_PER_SIMD=2, which left the register count and so the residency as they were), fresh-process A/B.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_48
mkdir -p $OUT
cd $REPO
fresh() {  # fresh <workload> <tag> <lib or "">
  env OTMB_LIB_OVERRIDE=$3 timeout -k 10 300 python3 bench.py --workload $1 --extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$2','workload':'$1','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
  rc=$?; if [ $rc -ne 0 ]; then echo "($1 $2: rc=$rc -- not measured)"; fi
}
D=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip.so
V=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_wg2.so
W=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_wg1.so
for r in 1 2 3; do
  fresh access1deg three_per_cu $D | tee -a $OUT/fresh_wps.jsonl
  fresh access1deg two_per_cu $V | tee -a $OUT/fresh_wps.jsonl
  fresh access1deg one_per_cu $W | tee -a $OUT/fresh_wps.jsonl
done
for r in 1 2; do
  fresh quarterdeg three_per_cu $D | tee -a $OUT/fresh_wps.jsonl
  fresh quarterdeg two_per_cu $V | tee -a $OUT/fresh_wps.jsonl
  fresh quarterdeg one_per_cu $W | tee -a $OUT/fresh_wps.jsonl
done
