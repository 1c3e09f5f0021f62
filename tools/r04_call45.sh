#!/bin/bash
# Round-4 GPU call 45: the adjusted tests under the look-back / dense switches, then the whole suite once more on the final sources (choose_placement now runs inside the 0.25 degree full-size test).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_45
mkdir -p $OUT
cd $REPO
for e in OTMB_LOOKBACK=1 OTMB_DENSE=1 OTMB_MARCH_ROWS=0; do
  env $e timeout -k 10 300 python3 -m pytest tests/test_formulations.py tests/test_bench_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_$e.log 2>&1; echo "$e rc=$? $(tail -1 $OUT/pytest_$e.log)"
done
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -3 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || { tail -40 $OUT/pytest_gpu.log; exit 1; }
echo "== done =="
