#!/bin/bash
# Round-4 GPU call 50: bench.py's multi-rank path on the one GPU (ranks share it, planes over gloo): --gpus 2 and --gpus 4, weak-scaled headline + the strong-scaled config4 leg.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_50
mkdir -p $OUT
cd $REPO
for n in 2 4; do
  OTMB_DIST_BACKEND=gloo OTMB_SHARE_GPU=1 timeout -k 10 500 python3 bench.py --gpus $n --steps 5 --warmup 2 --repeats 2 > $OUT/bench_gpus$n.json 2> $OUT/bench_gpus$n.err; rc=$?
  echo "--gpus $n rc=$rc"
  if [ $rc -ne 0 ]; then tail -15 $OUT/bench_gpus$n.err; exit 1; fi
  python3 -c "
import json; d=json.load(open('$OUT/bench_gpus$n.json'))
print(d['n_gpus'], d['scaling'], round(d['value']/1e9,3), round(d['ms_per_step'],3), d['config']['workload'][:110], d['config'].get('ranks_over'))
c=d.get('config4') or {}; print('config4:', c.get('error'), c.get('scaling'), c.get('grid'), c.get('ms_per_step'), c.get('wet_cells'))"
done
