#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then PMC passes.
# Usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the trace runs bench.py's default steps / warm-up / repeats (the fill pass needs ~20 steps to reach its steady rate: with
# --steps 10 --warmup 2 it measures 0.40-0.43 ms where the default run measures 0.38-0.39); the counter passes are short
TARGS="--no-cpu-baseline --no-end-to-end --extra-configs= $@"
ARGS="--steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs= $@"
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $TARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
# PMC_SETS=traffic: only the two HBM byte counters (the 0.1 degree grid: every pass takes minutes)
if [ "$PMC_SETS" = traffic ]; then
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
    echo "pmc [$set] rc=$?"
  done
else
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum" ; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
  echo "pmc [$set] rc=$?"
done
fi
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/pmc_*/ $OUT/trace   # (raw per-dispatch tables are large: the summaries travel back)
find $OUT -name "*kernel_stats.csv" | head -3
