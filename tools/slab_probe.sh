#!/bin/bash
# Same box, fresh processes alternating: the single-GPU path against the depth-slab path forced onto one rank
# (OTMB_FORCE_SLAB=1 under torchrun --nproc-per-node 1): what the slab orchestration costs per step, and whether the
# two ways of allocating the arrays draw different fill-pass times.  Writes gpurun_out/r05/slab_probe.jsonl.
set -e
mkdir -p gpurun_out/r05
out=gpurun_out/r05/slab_probe.jsonl
: > $out
port=29520
for r in 1 2 3 4; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-end-to-end --extra-configs= 2>/dev/null | python tools/jsonl_tag.py single $r >> $out
  port=$((port+1))
  OTMB_FORCE_SLAB=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline --no-end-to-end --extra-configs= 2>/dev/null | python tools/jsonl_tag.py forced_slab $r >> $out
done
python - <<'PY'
import json
for l in open("gpurun_out/r05/slab_probe.jsonl"):
    d = json.loads(l)
    print(d["tag"], d["round"], round(d["ms_per_step"], 4), d["kernels_ms"])
PY
