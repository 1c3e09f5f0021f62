#!/bin/bash
# Instruction-cache / scalar-data-cache / fetch counters of the hot kernels (gpurun).  Each pass is bounded.
TAG=${1:-ic}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQ_INSTS_SMEM SQ_INSTS_BRANCH" ; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$i.err
  echo "pmc [$set] rc=$?"
done
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
