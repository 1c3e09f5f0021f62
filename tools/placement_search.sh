#!/bin/bash
# Placement search (round 4): fresh bench.py processes at 1 degree (or $1) with the assembler's arrays carved out of one arena at controlled relative offsets.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-access1deg}; GB=${2:-8}; OUTF=${3:-$REPO/gpurun_out/placement_search_$WL.jsonl}
mkdir -p $(dirname $OUTF)
cd $REPO
run() {  # run <tag> ENV...
  tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload $WL --extra-configs= --no-cpu-baseline --no-end-to-end --warmup 3 --steps 10 --repeats 3 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$WL','ms_per_step':round(d['ms_per_step'],4),'fill':round(d['kernels_ms']['tm_kernel<fill>'],4),'ff':round(d['kernels_ms']['facefluxes_kernel'],4),'count':round(d['kernels_ms']['tm_count_kernel'],4),'frac':round(d['roofline']['frac'],4)}))" | tee -a $OUTF
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
}
M2=2097152
run separate X=0
run arena OTMB_ARENA_GB=$GB
run arena_align2m OTMB_ARENA_GB=$GB OTMB_ARENA_ALIGN=$M2
for pad in 256 512 1024 4096 16384 65536 262144 1048576; do
  run arena_pad$pad OTMB_ARENA_GB=$GB OTMB_ARENA_PAD=$pad
done
for pad in 4096 65536 1048576; do
  run arena_align2m_pad$pad OTMB_ARENA_GB=$GB OTMB_ARENA_ALIGN=$M2 OTMB_ARENA_PAD=$pad
done
run separate_align2m OTMB_ARENA_ALIGN=$M2
run separate_pad4096 OTMB_ARENA_PAD=4096
run separate_pad65536 OTMB_ARENA_PAD=65536
run separate X=0
run arena OTMB_ARENA_GB=$GB
echo "== done =="
