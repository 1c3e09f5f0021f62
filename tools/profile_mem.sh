#!/bin/bash
# Memory-system counters of the hot kernels (run on the GPU box via gpurun): address translation, L1 (TCP) stalls,
# texture addresser (TA), L2 (TCC) stalls towards DRAM.  One rocprofv3 --pmc pass per group; summary by pmc_summary.py.
# (A pass with TA_* counters -- TA_TA_BUSY_sum etc. -- hung on this pool and was removed; every pass is bounded.)
# Usage: tools/profile_mem.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-mem}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs= $@"
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCP_LATENCY_sum" \
           "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$i.err
  echo "pmc [$set] rc=$?"
done
python3 $REPO/tools/pmc_summary.py $OUT "tm_kernel,tm_count,facefluxes,dm_" > $OUT/pmc_summary.txt
rm -rf $OUT/pmc_*/   # (the raw per-dispatch tables are large: only the summary travels back)
