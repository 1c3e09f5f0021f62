#!/usr/bin/env python3
"""profiles/traffic.json from tools/profile.sh runs:
    python tools/make_traffic_json.py profiles/r03 access1deg=gpurun_out/prof_r03_1deg quarterdeg=gpurun_out/prof_r03_qdeg
Copies each run's rocprofv3 --stats kernel summary, PMC summary and bench line into profiles/<round>/ and writes the per-launch
HBM-side bytes of the hot kernels per workload, keyed to the hash of the kernel sources they were measured on.
Calibration (profiles/r03/README.md, tools/micro/stream_mix.hip on known byte counts): on gfx950 every TCC_EA0_RDREQ is a
128-byte request and FETCH_SIZE tallies it as 64 bytes -- for 8- and 16-byte-per-lane streams and for 8-byte gathers alike --
so fetch_bytes = 2 x FETCH_SIZE (= 128 x TCC_EA0_RDREQ); WRITE_SIZE is exact (64- and 32-byte requests)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dst = sys.argv[1]
os.makedirs(dst, exist_ok=True)
# kernel name patterns of the PMC summary -> the names bench.py uses (round 5: the fill pass is tm_kernel<1, 0>, the fused step's variants
# -- tm_kernel<1, 1 | 2> and the ϕtop-only facefluxes, whose last template argument is true -- get their own entries)
# (round 6: tm_kernel<FUSED, GIVEN> -- <0, 0> is the fill pass, <1 | 2, 0> the fused step's with Float64 / Float32 transports, <., 1> the same with
# a given TκH read where it lies instead of re-derived; GIVEN 2 / 3 -- a TκVdeep of another κ read too -- are not profiled)
names = {"tm_kernel<0, 0>": "tm_kernel<fill>", "tm_kernel<1, 0>": "tm_kernel<fill, fused>", "tm_kernel<2, 0>": "tm_kernel<fill, fused>",
         "tm_kernel<0, 1>": "tm_kernel<fill, TκH read>", "tm_kernel<1, 1>": "tm_kernel<fill, fused, TκH read>", "tm_kernel<2, 1>": "tm_kernel<fill, fused, TκH read>",
         "tm_count_kernel": "tm_count_kernel"}


def ff_name(k):
    if "facefluxes_kernel" not in k:
        return None
    return "facefluxes_kernel<top only>" if re.search(r",\s*true\s*>\s*$", k.strip()) and k.count(",") >= 5 else "facefluxes_kernel"
out = {
    "_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (tools/profile.sh = `bench.py --steps 10 --warmup 2`, separate passes for "
                "FETCH_SIZE and WRITE_SIZE).  gfx950 correction: fetch_bytes = 2 x FETCH_SIZE (every TCC_EA0_RDREQ is a 128-byte request "
                "tallied as 64 bytes; calibrated on 2 GiB streams of 8 and 16 bytes per lane and on an 8-byte gather, tools/micro/stream_mix.hip, "
                "profiles/r03/calibration_summary.txt); WRITE_SIZE is exact.  Infinity-Cache hits are counted as fetches.  kernel_source_sha16 = "
                "bench.kernel_source_hash() at the time of the measurement: bench.py reports traffic: null when the kernel sources have changed since.",
    "kernel_source_sha16": bench.kernel_source_hash(), "workloads": {},
}
for arg in sys.argv[2:]:
    wl, src = arg.split("=", 1)
    tag = os.path.join(dst, wl)
    for f, t in (("kernel_stats.csv", "_kernel_stats.csv"), ("pmc_summary.txt", "_pmc_summary.txt"), ("bench_trace.json", "_bench_under_rocprof.json")):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), tag + t)
    kern, cur = {}, None
    for line in open(os.path.join(src, "pmc_summary.txt")):
        if not line.startswith(" "):
            cur = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+mean\s+([\d.]+)", line)
            if m:
                kern.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    rec = {}
    for k, d in kern.items():
        cands = [name for pat, name in names.items() if pat in k]
        if ff_name(k):
            cands.append(ff_name(k))
        for name in cands:
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                fb, wb = 2 * 1024 * d["FETCH_SIZE"], 1024 * d["WRITE_SIZE"]
                rec[name] = dict(fetch_bytes=fb, write_bytes=wb, traffic_bytes=fb + wb, rdreq_128B=d.get("TCC_EA0_RDREQ_sum"),
                                 wrreq=d.get("TCC_EA0_WRREQ_sum"), wrreq_64B=d.get("TCC_EA0_WRREQ_64B_sum"))
    out["workloads"][wl] = rec
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["workloads"], indent=1))
