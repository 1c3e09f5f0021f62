#!/bin/bash
# The allocation lottery again (section 4 of profiles/r06/README.md excluded the HBM channels): does the fill pass's time per
# output set follow ADDRESS TRANSLATION -- misses of the CUs' UTCL1, stalls on the UTCL2 -- i.e. the physical fragment sizes behind a set's pages?
# One 0.25 degree process, four output sets, three launches each, one rocprofv3 pass per counter pair (--pmc with --kernel-trace only).
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/prof_r06_tlb
mkdir -p $OUT
REPO=$PWD
WL=${1:-quarterdeg}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_STALL_MULTI_MISS" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS"; do
  i=$((i + 1))
  rm -rf $OUT/raw_$i
  rocprofv3 --pmc $set --kernel-trace --output-format json -d $OUT/raw_$i -- python3 $REPO/tools/placement_channels.py $WL 4 3 $OUT/launches_$i.json > $OUT/run_$i.log 2>&1
  echo "pmc [$set] rc=$?"
  python3 $REPO/tools/tcc_channels.py $OUT/raw_$i "tm_kernel<0, 0>" --per-dispatch > $OUT/counters_$i.json 2> $OUT/counters_$i.err
  python3 - $OUT/raw_$i $OUT/kernel_times_$i.json <<'PY'
import glob, json, sys
root, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(root + "/**/*results.json", recursive=True):
    d = json.load(open(f)); tool = d.get("rocprofiler-sdk-tool", d); tool = tool[0] if isinstance(tool, list) else tool
    names = {ks.get("kernel_id"): (ks.get("formatted_kernel_name") or ks.get("kernel_name")) for ks in tool.get("kernel_symbols", [])}
    for r in tool.get("buffer_records", {}).get("kernel_dispatch", []):
        di = r.get("dispatch_info", {})
        if "tm_kernel<0, 0>" in (names.get(di.get("kernel_id"), "") or ""):
            rows.append({"start": r.get("start_timestamp"), "ms": (r.get("end_timestamp", 0) - r.get("start_timestamp", 0)) / 1e6})
rows.sort(key=lambda x: x["start"])
json.dump([round(x["ms"], 4) for x in rows], open(dst, "w"))
PY
  rm -rf $OUT/raw_$i
done
python3 - $OUT <<'PY'
import json, sys, glob, os
out = sys.argv[1]
table = {}
for f in sorted(glob.glob(os.path.join(out, "counters_*.json"))):
    i = f.split("_")[-1].split(".")[0]
    d = json.load(open(f)).get("per_dispatch", {})
    times = json.load(open(os.path.join(out, f"kernel_times_{i}.json")))
    for kname, cs in d.items():
        for nm, rows in cs.items():
            table[nm] = {"sum_per_dispatch": [r["sum"] for r in rows], "fill_ms_same_pass": times}
json.dump(table, open(os.path.join(out, "tlb_by_output_set.json"), "w"), indent=1)
for nm, t in table.items():
    print(nm, [f"{x:.4g}" for x in t["sum_per_dispatch"]], t["fill_ms_same_pass"])
PY
