#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile.sh run:  python tools/make_traffic_json.py gpurun_out/prof_<tag> profiles/r02/<name>
Copies the rocprofv3 --stats kernel summary and the PMC summary into profiles/ and writes the per-launch HBM-side bytes of
the three hot kernels, keyed to the hash of the kernel sources they were measured on (bench.kernel_source_hash)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst), exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], dst + "_kernel_stats.csv")
shutil.copy(os.path.join(src, "pmc_summary.txt"), dst + "_pmc_summary.txt")
if os.path.exists(os.path.join(src, "bench_trace.json")):
    shutil.copy(os.path.join(src, "bench_trace.json"), dst + "_bench_under_rocprof.json")
agg = {}
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
names = {"tm_kernel<1>": "tm_kernel<fill>", "tm_count_kernel": "tm_count_kernel", "facefluxes_kernel": "facefluxes_kernel"}
kernels = {}
for k, d in agg.items():
    for pat, name in names.items():
        if pat in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            fb = 1024 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
            wb = 1024 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            kernels[name] = dict(fetch_bytes=fb, write_bytes=wb, traffic_bytes=fb + wb)
out = {
    "_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in KiB; separate passes; tools/profile.sh = "
                "`bench.py --steps 10 --warmup 2`).  gfx950 correction: FETCH_SIZE was calibrated at 1.0x for this access pattern (8-byte-per-lane "
                "loads) on facefluxes_kernel, whose read set is known exactly (umo+vmo+wet3D + neighbour-row re-reads); WRITE_SIZE is exact.  "
                "kernel_source_sha16 = bench.kernel_source_hash() at the time of the measurement: bench.py reports traffic: null when the "
                "kernel sources have changed since.",
    "workload": "access1deg", "source": os.path.relpath(dst + "_pmc_summary.txt", ROOT), "kernel_source_sha16": bench.kernel_source_hash(),
    "kernels": kernels,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
