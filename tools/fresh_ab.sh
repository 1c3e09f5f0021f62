#!/bin/bash
# A/B of library builds in FRESH processes (what bench.py and the driver see): tools/fresh_ab.sh <workload> <rounds> <lib name or "default"> ...
WL=$1; R=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for r in $(seq 1 $R); do
  for name in "$@"; do
    if [ "$name" = default ]; then L=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip.so; else L=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_$name.so; fi
    OTMB_LIB_OVERRIDE=$L python3 bench.py --workload $WL --extra-configs "" --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('$name', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms'].items()}, round(d['roofline']['frac'],3))"
  done
done
