#!/usr/bin/env python3
"""After `tools/r06_profiles.sh` on the GPU box: copy the four runs into profiles/r06/, rewrite profiles/traffic.json for the current kernel
sources and the table of profiles/r06/README.md section 1 from the copied files (so that the README quotes what the directory holds).
    python3 tools/r06_refresh_profiles.py <call number>"""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
call = sys.argv[1]
runs = {"access1deg": "prof_r06_1deg", "quarterdeg": "prof_r06_qdeg", "access1deg_given": "prof_r06_1deg_given", "quarterdeg_given": "prof_r06_qdeg_given"}
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), os.path.join(ROOT, "profiles", "r06"),
                       *[f"{k}={os.path.join(ROOT, 'gpurun_out', v)}" for k, v in runs.items()]], stdout=subprocess.DEVNULL)


def row(f, pat):
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06", f"{f}_kernel_stats.csv"))):
        if pat in r["Name"]:
            return float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3
    raise KeyError((f, pat))


def bench_ms(f):
    d = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r06", f"{f}_bench_under_rocprof.json")) if l.startswith("{")][-1])
    return d["roofline"]["avg_kernel_ms"], d["roofline"]["algorithmic_bytes_per_launch"] / 1e9


t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["workloads"]
A, G, QA, QG = row("access1deg", "tm_kernel<0, 0>"), row("access1deg_given", "tm_kernel<0, 1>"), row("quarterdeg", "tm_kernel<0, 0>"), row("quarterdeg_given", "tm_kernel<0, 1>")
FA, FG, FQA, FQG = row("access1deg", "tm_kernel<1, 0>"), row("access1deg_given", "tm_kernel<1, 1>"), row("quarterdeg", "tm_kernel<1, 0>"), row("quarterdeg_given", "tm_kernel<1, 1>")
(ba, a0), (bg, a1), (bqa, a2), (bqg, a3) = bench_ms("access1deg"), bench_ms("access1deg_given"), bench_ms("quarterdeg"), bench_ms("quarterdeg_given")
tr = lambda wl, k: (t[wl][k]["fetch_bytes"] / 1e9, t[wl][k]["write_bytes"] / 1e9)
ta, tg, tqa, tqg = tr("access1deg", "tm_kernel<fill>"), tr("access1deg_given", "tm_kernel<fill, TκH read>"), tr("quarterdeg", "tm_kernel<fill>"), tr("quarterdeg_given", "tm_kernel<fill, TκH read>")
new = f'''| `tm_kernel<0, GIVEN>` (fill; call {call}, one box -- the pool's boxes differ: calls 21 / 25 / 26 / 30 measured 343 / 325 / 335 / 386 µs and 260 / 247 / 247 / 278 µs with the same kernels (call 30: the slowest box of the round, 0.25 degree 6.75 ms)) | all five matrices (`<0, 0>`) | TκH + TκVdeep given (`<0, 1>`: TκH read) |
|---|---|---|
| 1 degree: average / min of 123 launches | {A[0]:.1f} / {A[1]:.1f} µs | {G[0]:.1f} / {G[1]:.1f} µs |
| 1 degree: algorithmic bytes -> GB/s, of 8 TB/s | {a0:.4f} GB -> {a0 / A[0] * 1e6:,.0f} GB/s = **{a0 / A[0] * 1e6 / 8000:.3f}** | {a1:.4f} GB -> {a1 / G[0] * 1e6:,.0f} GB/s = **{a1 / G[0] * 1e6 / 8000:.3f}** |
| 1 degree: HBM traffic (2 x FETCH_SIZE + WRITE_SIZE) | {ta[0]:.3f} + {ta[1]:.3f} = {sum(ta):.3f} GB ({sum(ta) / a0:.2f} x) | {tg[0]:.3f} + {tg[1]:.3f} = {sum(tg):.3f} GB ({sum(tg) / a1:.2f} x: the given TκH's values and offsets are read, 0.108 + 0.024 GB, `thkcello` is not) |
| 0.25 degree: average (as allocated: no placement choice under a profiler) | {QA[0] / 1e3:.3f} ms -> {a2 / QA[0] * 1e6:,.0f} GB/s = **{a2 / QA[0] * 1e6 / 8000:.3f}** | {QG[0] / 1e3:.3f} ms (min {QG[1] / 1e3:.3f}) -> {a3 / QG[0] * 1e6:,.0f} GB/s = {a3 / QG[0] * 1e6 / 8000:.3f} |
| 0.25 degree: HBM traffic | {tqa[0]:.2f} + {tqa[1]:.2f} = {sum(tqa):.2f} GB ({sum(tqa) / a2:.2f} x) | {tqg[0]:.2f} + {tqg[1]:.2f} = {sum(tqg):.2f} GB ({sum(tqg) / a3:.2f} x) |
| fused step's fill (`tm_kernel<1, .>`), 1 degree / 0.25 degree | {FA[0]:.1f} µs / {FQA[0] / 1e3:.3f} ms | {FG[0]:.1f} µs / {FQG[0] / 1e3:.3f} ms |
| the same command's HIP-event average in `bench.py` (`*_bench_under_rocprof.json`) | {ba * 1e3:.2f} µs / {bqa:.3f} ms | {bg * 1e3:.2f} µs / {bqg:.3f} ms |

'''
p = os.path.join(ROOT, "profiles", "r06", "README.md")
s = open(p, encoding="utf-8").read()
a = s.index("| `tm_kernel<0, GIVEN>` (fill; call ")
b = s.index("The written bytes fall by 41 %")
open(p, "w", encoding="utf-8").write(s[:a] + new + s[b:])
sys.path.insert(0, ROOT)
import bench

print(new)
print("kernel sources", bench.kernel_source_hash(), "traffic.json", json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["kernel_source_sha16"])
