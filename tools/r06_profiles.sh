#!/bin/bash
# Round 6 profile runs (on the GPU box, via gpurun): rocprofv3 --kernel-trace --stats + the two HBM byte counters of bench.py at 1 and
# 0.25 degree, for the full build and for the given-operators path (--given-ops).  Summaries land in gpurun_out/prof_r06_*.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PMC_SETS=traffic
bash tools/profile.sh r06_1deg --workload access1deg 2>&1 | tail -3
bash tools/profile.sh r06_1deg_given --workload access1deg --given-ops 2>&1 | tail -3
bash tools/profile.sh r06_qdeg --workload quarterdeg 2>&1 | tail -3
bash tools/profile.sh r06_qdeg_given --workload quarterdeg --given-ops 2>&1 | tail -3
ls gpurun_out/prof_r06_*
