#!/bin/bash
# Round-4 GPU call 13: does the headline measure slower when its process starts AFTER the two big child processes (bench.py's default order)?
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_13
mkdir -p $OUT
cd $REPO
one() {  # one <tag> <extra-configs>
  timeout -k 10 700 python3 bench.py --no-cpu-baseline --no-end-to-end --extra-configs "$2" 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$1','ms_per_step':round(d['ms_per_step'],4),'fill':round(d['kernels_ms']['tm_kernel<fill>'],4),'ff':round(d['kernels_ms']['facefluxes_kernel'],4),'frac':round(d['roofline']['frac'],4),
 'config3_fill': (d.get('config3') or {}).get('kernels_ms',{}).get('tm_kernel<fill>'), 'config5_fill': (d.get('config5') or {}).get('kernels_ms',{}).get('tm_kernel<fill>')}))" | tee -a $OUT/order.jsonl
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo STOP; exit 1; fi
}
for r in 1 2; do
  one standalone ""
  one after_children "quarterdeg,tenthdeg"
  one standalone ""
  one after_quarterdeg_only "quarterdeg"
done
echo "== done =="
