#!/bin/bash
# Round-4 GPU call 44: the whole GPU suite under every alternative code path of the library (environment switches read when a context is created), final sources.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_44
mkdir -p $OUT
cd $REPO
run() {  # run <tag> ENV...
  tag=$1; shift
  env "$@" timeout -k 10 600 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_$tag.log 2>&1; rc=$?
  echo "$tag ($*): rc=$rc  $(tail -1 $OUT/pytest_$tag.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "STOP (timeout)"; exit 1; fi
  if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR" $OUT/pytest_$tag.log | head -10; fi
}
run lookback OTMB_LOOKBACK=1
run dense OTMB_DENSE=1
run ff_rows4 OTMB_FF_ROWS=4
run ff_rows4_no_lds OTMB_FF_ROWS=4 OTMB_FF_LDS_SOUTH=0
run count_order0_ff_blockidx OTMB_COUNT_ORDER=0 OTMB_FF_XCD=0 OTMB_DEAL_HEAVY=0
run march_cols37_rows3 OTMB_MARCH_COLS=37 OTMB_MARCH_ROWS=3
run wet_rank_order OTMB_MARCH_ROWS=0
echo "== done =="
