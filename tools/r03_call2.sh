#!/bin/bash
# Round-3 GPU call 2: write-rate shapes, torch's own fill / copy on the same box, the new asynchronous-pipeline tests.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_2
mkdir -p $OUT
cd $REPO
timeout -k 10 300 python3 -m pytest tests/test_pipeline_errors.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || { echo "tests failed rc=$rc"; tail -40 $OUT/pytest.log; exit 1; }
timeout -k 10 120 tools/micro/stream_mix > $OUT/stream_mix.log 2>&1 || exit 1
cat $OUT/stream_mix.log
timeout -k 10 120 python3 tools/membw.py > $OUT/membw.log 2>&1 || exit 1
cat $OUT/membw.log
