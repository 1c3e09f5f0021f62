#!/bin/bash
# SQ-side memory-instruction counters of the hot kernels (gpurun).  Each pass is bounded: a TA_* counter pass hung once.
TAG=${1:-sq}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline"
i=0
for set in "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES SQ_WAVE_CYCLES" ; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$i.err
  echo "pmc [$set] rc=$?"
done
