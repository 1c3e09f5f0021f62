#!/bin/bash
# Round-4 GPU call 25: march order in blocks of columns (OTMB_MARCH_COLS): parity, then fresh-process A/B on the 0.1 / 0.25 / 1 degree grids, then the
# HBM fetch of the fill and counting passes at 0.1 degree with the new default.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_25
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python3 -m pytest tests/test_formulations.py -m gpu -q -x > $OUT/pytest_formulations.log 2>&1; rc=$?
tail -3 $OUT/pytest_formulations.log
if [ $rc -ne 0 ]; then echo "STOP tests rc=$rc"; exit 1; fi
fresh() {  # fresh <workload> <tag> <steps> ENV...
  wl=$1; tag=$2; st=$3; shift; shift; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload $wl --extra-configs= --no-cpu-baseline --no-end-to-end --steps $st --warmup 2 --repeats 2 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
  rc=$?; if [ $rc -ne 0 ]; then echo "STOP $wl $tag rc=$rc"; exit 1; fi
}
for r in 1 2; do
  fresh tenthdeg whole_rows 4 OTMB_MARCH_COLS=0 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg auto_1200 4 OTMB_MARCH_COLS=-1 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg cols_900 4 OTMB_MARCH_COLS=900 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg cols_1800 4 OTMB_MARCH_COLS=1800 | tee -a $OUT/fresh_tenthdeg.jsonl
  fresh tenthdeg cols_600 4 OTMB_MARCH_COLS=600 | tee -a $OUT/fresh_tenthdeg.jsonl
done
fresh tenthdeg rows4_whole 4 OTMB_MARCH_COLS=0 OTMB_MARCH_ROWS=4 | tee -a $OUT/fresh_tenthdeg.jsonl
fresh tenthdeg rows16_1200 4 OTMB_MARCH_ROWS=16 | tee -a $OUT/fresh_tenthdeg.jsonl
fresh tenthdeg rows4_1200 4 OTMB_MARCH_ROWS=4 | tee -a $OUT/fresh_tenthdeg.jsonl
for r in 1 2; do
  fresh quarterdeg whole_rows 10 OTMB_MARCH_COLS=0 | tee -a $OUT/fresh_quarterdeg.jsonl
  fresh quarterdeg cols_720 10 OTMB_MARCH_COLS=720 | tee -a $OUT/fresh_quarterdeg.jsonl
  fresh quarterdeg cols_480 10 OTMB_MARCH_COLS=480 | tee -a $OUT/fresh_quarterdeg.jsonl
done
for r in 1 2; do
  fresh access1deg whole_rows 10 OTMB_MARCH_COLS=0 | tee -a $OUT/fresh_access1deg.jsonl
  fresh access1deg cols_180 10 OTMB_MARCH_COLS=180 | tee -a $OUT/fresh_access1deg.jsonl
done
cd /tmp && export TMPDIR=/tmp
ARGS="--workload tenthdeg --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs="
P=$REPO/gpurun_out/prof_r04_tenthdeg_cols
mkdir -p $P
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/pmc_$set -- python3 $REPO/bench.py $ARGS > /dev/null 2> $P/pmc_$set.err
  rc=$?; echo "pmc [$set] rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $P/pmc_$set.err; exit 1; fi
done
python3 $REPO/tools/pmc_summary.py $P "tm_kernel,tm_count,facefluxes" | tee $P/pmc_summary.txt
rm -rf $P/pmc_*/
echo "== done =="
