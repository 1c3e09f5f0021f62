#!/bin/bash
# Round-4 GPU call 24: the 0.1 degree grid (BASELINE config 5's grid on ONE GPU) under rocprofv3: kernel trace + stats, then HBM traffic (one counter per pass).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r04_tenthdeg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload tenthdeg --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-end-to-end --extra-configs="
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rc=$?; echo "trace rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $OUT/trace.err; exit 1; fi
cat $OUT/bench_trace.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['kernels_ms'], d['roofline'])"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
  rc=$?; echo "pmc [$set] rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $OUT/pmc_$name.err; exit 1; fi
done
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/pmc_*/ $OUT/trace
cat $OUT/kernel_stats.csv | head -12
cat $OUT/pmc_summary.txt
echo "== done =="
