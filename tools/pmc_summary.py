#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv files: mean counter value per kernel per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
keep = sys.argv[2].split(",") if len(sys.argv) > 2 else ("tm_kernel", "tm_count", "facefluxes", "tilescan", "push_mask", "order_")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if not any(t in k for t in keep): continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
