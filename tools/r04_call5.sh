#!/bin/bash
# Round-4 GPU call 5: the round's profiles with the final kernels -- rocprofv3 kernel trace + stats and counter passes of bench.py at 1 and 0.25 degree
# (tools/profile.sh), memory-system counters (tools/profile_mem.sh), and one default bench.py run (the driver's command).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_5
mkdir -p $OUT
cd $REPO
echo "== default bench run =="
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?
echo "bench rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','kernels_ms','step_gbs')}); print(d['roofline']); print(d.get('config2')); print(d.get('box_probe'))
print({k:(d[k].get('ms_per_step'), d[k].get('roofline',{}).get('frac') if d[k].get('roofline') else None, d[k].get('kernels_ms')) for k in ('config3','config5') if k in d})
print(d.get('end_to_end')); print({k:v for k,v in (d.get('cpu_baseline') or {}).items() if k!='sample'})"
echo "== profile 1 degree =="
timeout -k 10 900 bash tools/profile.sh r04_1deg > $OUT/profile_1deg.log 2>&1; rc=$?; tail -3 $OUT/profile_1deg.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
echo "== profile 0.25 degree =="
timeout -k 10 1100 bash tools/profile.sh r04_qdeg --workload quarterdeg > $OUT/profile_qdeg.log 2>&1; rc=$?; tail -3 $OUT/profile_qdeg.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
echo "== memory counters =="
timeout -k 10 900 bash tools/profile_mem.sh r04_1deg_mem > $OUT/mem_1deg.log 2>&1; rc=$?; tail -2 $OUT/mem_1deg.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
timeout -k 10 1100 bash tools/profile_mem.sh r04_qdeg_mem --workload quarterdeg > $OUT/mem_qdeg.log 2>&1; rc=$?; tail -2 $OUT/mem_qdeg.log
echo "== done =="
