#!/bin/bash
# Fresh processes alternating on one box: tools/host_ab.py under the variants given as "NAME:ENV=VAL,ENV=VAL" arguments ("base:" = no switch).
# Usage (GPU box):  bash tools/host_ab.sh <tag> <rounds> base: narrow1:OTMB_XFER_NARROW=1 noshift:OTMB_FF_SHIFT_ON_HOST=0
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/${TAG}_host_ab.jsonl
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; python3 tools/host_ab.py 2>/dev/null | sed "s/^{/{\"variant\": \"$name\", \"round\": $r, /" >> $OUT )
  done
done
cat $OUT | cut -c1-600
