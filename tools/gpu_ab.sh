#!/bin/bash
# ONE runner for every fresh-process A/B on the GPU box (replaces the per-call r03_/r04_callNN.sh scripts):
#   tools/gpu_ab.sh <out tag> <rounds> <workload> <steps> <variant> [<variant> ...]
# variant = name[:ENV=value[,ENV=value...]]   -- the variants alternate inside every round (box drift hits them alike);
#   a variant whose ENV list contains LIB=<name> runs the library build lib/libotmb_hip_<name>.so (tools/_build_variant.py).
# One JSON line per run goes to gpurun_out/r05/<out tag>.jsonl: {tag, workload, ms_per_step, kernels_ms, frac}.
# Extra bench.py flags: BENCH_FLAGS="--protocol twophase" tools/gpu_ab.sh ...
set -o pipefail
TAG=$1; ROUNDS=$2; WL=$3; STEPS=$4; shift 4
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${OTMB_AB_DIR:-r06}
mkdir -p "$OUT"
cd "$REPO" || exit 1
for r in $(seq 1 "$ROUNDS"); do
  for v in "$@"; do
    name=${v%%:*}; envs=""
    if [ "$v" != "$name" ]; then envs=${v#*:}; fi
    args=()
    IFS=',' read -ra kv <<< "$envs"
    for e in "${kv[@]}"; do
      [ -z "$e" ] && continue
      if [ "${e%%=*}" = LIB ]; then args+=("OTMB_LIB_OVERRIDE=$REPO/oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_${e#*=}.so"); else args+=("$e"); fi
    done
    env "${args[@]}" timeout -k 10 600 python3 bench.py --workload "$WL" --extra-configs= --no-cpu-baseline --no-end-to-end --steps "$STEPS" --warmup 3 \
        --repeats 3 --placement-candidates 1 $BENCH_FLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print(json.dumps({'tag':'$name','round':$r,'workload':'$WL','ms_per_step':round(d['ms_per_step'],4),'kernels_ms':{k:round(x,4) for k,x in d['kernels_ms'].items()},'frac':round(d['roofline']['frac'],4),'fused':({'ms_per_step':round(d['fused_step']['ms_per_step'],4),'kernels_ms':{k:round(x,4) for k,x in d['fused_step']['kernels_ms'].items()}} if isinstance(d.get('fused_step'),dict) and 'ms_per_step' in d['fused_step'] else d.get('fused_step')),'box':{k:(round(x,4) if isinstance(x,float) else x) for k,x in (d['roofline'].get('box') or {}).items() if isinstance(x,(float,dict))}}))" | tee -a "$OUT/$TAG.jsonl"
    rc=$?; if [ $rc -ne 0 ]; then echo "($WL $name rc=$rc)" | tee -a "$OUT/$TAG.jsonl"; fi
  done
done
