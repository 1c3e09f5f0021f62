#!/bin/bash
# VERDICT r05 item 5 on the GPU box: per-channel HBM requests of the fill pass for four output sets of ONE process (0.25 degree).
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/prof_r06_channels
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "TCC_EA0_WRREQ" "TCC_EA0_RDREQ"; do
  rm -rf $OUT/raw_$set
  rocprofv3 --pmc $set --kernel-trace --output-format json -d $OUT/raw_$set -- python3 $REPO/tools/placement_channels.py quarterdeg 4 3 $OUT/launches_$set.json > $OUT/run_$set.log 2>&1
  echo "pmc [$set] rc=$?"
  python3 $REPO/tools/tcc_channels.py $OUT/raw_$set "tm_kernel<0>" --per-dispatch > $OUT/channels_$set.json 2> $OUT/channels_$set.err
  python3 - $OUT/raw_$set $OUT/kernel_times_$set.json <<'PY'
import glob, json, sys
# kernel durations of the same run (rocprofv3's own timestamps)
root, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(root + "/**/*results.json", recursive=True):
    d = json.load(open(f)); tool = d.get("rocprofiler-sdk-tool", d); tool = tool[0] if isinstance(tool, list) else tool
    names = {ks.get("kernel_id"): (ks.get("formatted_kernel_name") or ks.get("kernel_name")) for ks in tool.get("kernel_symbols", [])}
    for r in tool.get("buffer_records", {}).get("kernel_dispatch", []):
        di = r.get("dispatch_info", {})
        nm = names.get(di.get("kernel_id"), "")
        if "tm_kernel<0>" in (nm or ""):
            rows.append({"start": r.get("start_timestamp"), "ms": (r.get("end_timestamp", 0) - r.get("start_timestamp", 0)) / 1e6})
rows.sort(key=lambda x: x["start"])
json.dump([round(x["ms"], 4) for x in rows], open(dst, "w"))
PY
  rm -rf $OUT/raw_$set
done
ls -la $OUT
