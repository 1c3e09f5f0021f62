#!/bin/bash
# Placement search 2 (round 4): which side matters (inputs or outputs in one allocation), and does a coarse stride between the arrays of an arena change anything?
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
WL=access1deg; OUTF=$REPO/gpurun_out/placement_search2_$WL.jsonl
mkdir -p $(dirname $OUTF)
cd $REPO
run() {  # run <tag> ENV...
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload $WL --extra-configs= --no-cpu-baseline --no-end-to-end --warmup 3 --steps 10 --repeats 3 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','ms_per_step':round(d['ms_per_step'],4),'fill':round(d['kernels_ms']['tm_kernel<fill>'],4),'ff':round(d['kernels_ms']['facefluxes_kernel'],4),'count':round(d['kernels_ms']['tm_count_kernel'],4),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUTF
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
}
M=1048576
run separate X=0
run arena_inputs_only OTMB_ARENA_GB=0.85
run arena_outputs_only OTMB_ARENA_GB=3 OTMB_ARENA_WHEN=outputs
run arena_all OTMB_ARENA_GB=8
for mb in 2 4 6 8 10 12 14 16 18 22 26 30 32 34 46 62 64 66; do
  run arena_align2m_pad${mb}MB OTMB_ARENA_GB=12 OTMB_ARENA_ALIGN=$((2*M)) OTMB_ARENA_PAD=$((mb*M))
done
run separate X=0
echo "== done =="
