#!/bin/bash
# Round-4 final check: what the driver runs at round end -- pytest -m gpu, smoke(), python bench.py.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_final3
mkdir -p $OUT
cd $REPO
timeout -k 10 800 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -4 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?
echo "bench rc=$rc"
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','kernels_ms','step_gbs')}); print(d['roofline'])
print({k:(d[k].get('ms_per_step'), (d[k].get('roofline') or {}).get('frac')) for k in ('config3','config5') if k in d}); print((d.get('config2') or {}).get('ms'))
print({k:v for k,v in (d.get('cpu_baseline') or {}).items() if k not in ('sample','multithread')})"
echo "== done =="
