#!/bin/bash
# Memory-system counters of tools/micro/fill_model (loads only / stores only / both), the same groups as tools/profile_mem.sh:
# what a CU's loads and stores do to each other when the arithmetic is trivial.   tools/profile_fill_model.sh <tag> [fill_model args]
set -o pipefail
TAG=${1:-fm}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" ; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- $REPO/tools/micro/fill_model "$@" > $OUT/run_$i.log 2> $OUT/pmc_$i.err
  rc=$?; echo "pmc [$set] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 $REPO/tools/pmc_summary.py $OUT "plain,march" > $OUT/pmc_summary.txt
rm -rf $OUT/pmc_*/
