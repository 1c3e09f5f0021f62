#!/bin/bash
# Round-4 GPU call 12: the library's placement-aware device arrays (otmb_dev_alloc: small physical handles) in the device-resident pipeline:
# parity, then fresh-process A/B against torch's allocator.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_12
mkdir -p $OUT
cd $REPO
stop() { echo "STOP: $1 (rc=$2)"; exit 1; }
timeout -k 10 120 python3 -c "
import torch, numpy as np
from otmb_amd import capi
c = capi.Context(0)
t = c.dev_empty(1000003, torch.float64); t.fill_(3.0); print('dev_empty ok', t.sum().item(), t.data_ptr() % (2<<20), capi.lib().otmb_dev_alloc_mode())
u = c.dev_empty(5, torch.int64); u.copy_(torch.arange(5)); print(u.cpu().numpy())
del t, u
import gc; gc.collect(); print('freed')
" 2>&1 | tail -5
echo "== gpu tests =="
timeout -k 10 800 python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -4 $OUT/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop "gpu tests" $rc; fi
BARGS="--extra-configs= --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --repeats 3"
fresh() {  # fresh <workload> <tag> ENV...
  wl=$1; tag=$2; shift; shift
  env "$@" timeout -k 10 240 python3 bench.py --workload $wl $BARGS 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps({'tag':'$tag','workload':'$wl','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(v,4) for k,v in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4)}))"
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then stop fresh $rc; fi
}
for r in 1 2 3 4; do
  fresh access1deg torch OTMB_DEV_ARRAYS=torch | tee -a $OUT/fresh_devalloc.jsonl
  fresh access1deg otmb_2MB OTMB_DEV_ARRAYS=otmb | tee -a $OUT/fresh_devalloc.jsonl
done
fresh access1deg otmb_32MB OTMB_DEV_ALLOC_HANDLE_MB=32 | tee -a $OUT/fresh_devalloc.jsonl
fresh access1deg otmb_8MB OTMB_DEV_ALLOC_HANDLE_MB=8 | tee -a $OUT/fresh_devalloc.jsonl
fresh access1deg otmb_malloc OTMB_DEV_ALLOC=malloc | tee -a $OUT/fresh_devalloc.jsonl
for r in 1 2 3; do
  fresh quarterdeg torch OTMB_DEV_ARRAYS=torch | tee -a $OUT/fresh_devalloc.jsonl
  fresh quarterdeg otmb OTMB_DEV_ARRAYS=otmb | tee -a $OUT/fresh_devalloc.jsonl
done
fresh quarterdeg otmb_2MB OTMB_DEV_ALLOC_HANDLE_MB=2 | tee -a $OUT/fresh_devalloc.jsonl
echo "== done =="
