#!/bin/bash
# Round-4 GPU call 29: the round's final profiles on the final sources, three grid sizes: rocprofv3 kernel trace + stats, then HBM traffic and request counters (tools/profile.sh;
# for the 0.1 degree grid the shorter tools/r04_call24.sh sequence), then `python bench.py` as the driver runs it.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout -k 10 500 bash tools/profile.sh r04c_1deg > gpurun_out/prof_r04c_1deg.log 2>&1; echo "1deg rc=$?"; tail -3 gpurun_out/prof_r04c_1deg.log
timeout -k 10 700 bash tools/profile.sh r04c_qdeg --workload quarterdeg > gpurun_out/prof_r04c_qdeg.log 2>&1; echo "qdeg rc=$?"; tail -3 gpurun_out/prof_r04c_qdeg.log
rm -rf gpurun_out/prof_r04_tenthdeg
timeout -k 10 900 bash tools/r04_call24.sh > gpurun_out/prof_r04c_tenthdeg.log 2>&1; echo "tenthdeg rc=$?"; grep -E "^trace rc|^pmc" gpurun_out/prof_r04c_tenthdeg.log
mkdir -p gpurun_out/r04_29
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_29/bench_default.json 2> gpurun_out/r04_29/bench_default.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r04_29/bench_default.json'))
print({k: d[k] for k in ('value','ms_per_step','kernels_ms','step_gbs')}); print(d['roofline'])
print({k:(d[k].get('ms_per_step'), (d[k].get('roofline') or {}).get('frac')) for k in ('config3','config5') if k in d}); print((d.get('config2') or {}).get('ms'))
print({k:v for k,v in (d.get('end_to_end') or {}).items() if k != 'note'})
print({k:v for k,v in (d.get('cpu_baseline') or {}).items() if k not in ('sample','multithread')})"
echo "== done =="
