#!/bin/bash
# NOTE (round 6): needs the dense-march formulation, which left the product: apply tools/experiments/r06_removed_formulations.patch first.
# SQ counters of the transportmatrix kernels for one formulation: tools/profile_sq_dense.sh <tag> <OTMB_DENSE value> [bench args]
set -o pipefail
TAG=$1; DENSE=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs= $@"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT" ; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  OTMB_DENSE=$DENSE timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
  rc=$?; echo "pmc [$set] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 $REPO/tools/pmc_summary.py $OUT "tm_kernel,tm_count,dm_" > $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
rm -rf $OUT/pmc_*/
