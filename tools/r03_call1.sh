#!/bin/bash
# Round-3 GPU call 1: parity under march order, tile-order scan, stream ceiling, traffic counters at 0.25 degree.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_1
mkdir -p $OUT
cd $REPO
stop() { echo "STOP: $1 (rc=$2)"; exit 1; }
guard() { rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop "$1" $rc; fi; }

echo "== gpu tests with the march order forced (OTMB_MARCH_ROWS=2) =="
OTMB_MARCH_ROWS=2 timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_march2.log 2>&1; rc=$?
tail -3 $OUT/pytest_march2.log
[ $rc -eq 0 ] || stop "gpu tests under march order" $rc

echo "== tile-order scan =="
timeout -k 10 300 python3 tools/march_scan.py --workload quarterdeg --rows 0,2,4,8,16,32,64 --rounds 2 --steps 8 > $OUT/march_quarterdeg.jsonl 2> $OUT/march_quarterdeg.err; guard "march_scan quarterdeg"
cat $OUT/march_quarterdeg.jsonl
timeout -k 10 200 python3 tools/march_scan.py --workload access1deg --rows 0,2,4,8,16,32 --rounds 3 --steps 20 > $OUT/march_access1deg.jsonl 2> $OUT/march_access1deg.err; guard "march_scan access1deg"
cat $OUT/march_access1deg.jsonl

echo "== stream ceiling =="
timeout -k 10 120 tools/micro/stream_mix > $OUT/stream_mix.log 2>&1; guard "stream_mix"
cat $OUT/stream_mix.log

cd /tmp && export TMPDIR=/tmp
echo "== counter calibration on stream_mix =="
for set in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-30)
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/calib/pmc_$name -- $REPO/tools/micro/stream_mix > /dev/null 2> $OUT/calib_$name.err; guard "calib $set"
done
python3 $REPO/tools/pmc_summary.py $OUT/calib calib_,mix_stream > $OUT/calib_summary.txt
cat $OUT/calib_summary.txt

echo "== traffic of the fill pass at 0.25 degree: wet-rank order against march order =="
ARGS="--workload quarterdeg --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs="
for rows in 0 8; do
  for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-30)
    OTMB_MARCH_ROWS=$rows timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/q_rows$rows/pmc_$name -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/q_rows${rows}_$name.err; guard "pmc $set rows=$rows"
  done
  python3 $REPO/tools/pmc_summary.py $OUT/q_rows$rows > $OUT/q_rows${rows}_summary.txt
  echo "--- rows=$rows"; cat $OUT/q_rows${rows}_summary.txt
done
echo "== done =="
