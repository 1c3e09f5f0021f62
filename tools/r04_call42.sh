#!/bin/bash
# Round-4 GPU call 42: the final tree: supporting-kernel table (with makegridmetrics and the B-grid interpolation), whole GPU suite, smoke(), python bench.py.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_42
mkdir -p $OUT
cd $REPO
timeout -k 10 400 python3 tools/secondary_time.py access1deg > $OUT/secondary_access1deg.jsonl 2> $OUT/err.log; rc=$?
python3 -c "
import json
for l in open('$OUT/secondary_access1deg.jsonl'):
    d=json.loads(l)
    if 'call' in d: print(d['call'], d['ms'], d['algorithmic_MB'], d['frac_of_8TBps'])"
echo "secondary rc=$rc"; [ $rc -eq 0 ] || tail -8 $OUT/err.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -3 $OUT/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "STOP tests rc=$rc"; tail -40 $OUT/pytest_gpu.log; exit 1; fi
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?
echo "bench rc=$rc"
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','kernels_ms','step_gbs')}); print(d['roofline']); print(d.get('placement'))
print({k:(d[k].get('ms_per_step'), (d[k].get('roofline') or {}).get('frac'), (d[k].get('placement') or {}).get('fill_ms')) for k in ('config3','config5') if k in d}); print((d.get('config2') or {}).get('ms'))
print((d.get('box_probe') or {}).get('rocm_smi'))"
echo "== done =="
