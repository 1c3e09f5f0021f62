#!/bin/bash
# Round-4 GPU call 36: the profiles of the final sources once more (1 and 0.25 degree; call 29's profiled processes had drawn slow placements), bench.py's placement choice active.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout -k 10 500 bash tools/profile.sh r04d_1deg > gpurun_out/prof_r04d_1deg.log 2>&1; echo "1deg rc=$?"
timeout -k 10 700 bash tools/profile.sh r04d_qdeg --workload quarterdeg > gpurun_out/prof_r04d_qdeg.log 2>&1; echo "qdeg rc=$?"
for d in 1deg qdeg; do head -4 gpurun_out/prof_r04d_$d/kernel_stats.csv | cut -c1-110; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r04d_$d/bench_trace.json')); print(d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d.get('placement'), (d.get('box_probe') or {}).get('rocm_smi'))"; done
