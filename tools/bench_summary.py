#!/usr/bin/env python3
"""Prints the numbers of a bench.py JSON line that a reader compares between runs (notes dropped).  python tools/bench_summary.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def show(x, drop=("note", "what", "no_promise", "sample", "expected_step_note")):
    if isinstance(x, dict):
        return {k: show(v) for k, v in x.items() if k not in drop}
    if isinstance(x, float):
        return round(x, 5)
    return x


print("headline", show({k: d[k] for k in ("value", "ms_per_step", "repeats")}))
print("kernels", show(d["kernels_ms"]))
print("roofline", show({k: v for k, v in d["roofline"].items() if k != "box"}))
print("box", show(d["roofline"].get("box")))
for k in ("fused_step", "given_ops_step", "end_to_end", "cpu_baseline"):
    print(k, show(d.get(k)))
for c in ("config3", "config5"):
    if c in d:
        print(c, show({k: v for k, v in d[c].items() if k in ("ms_per_step", "value", "error", "skipped", "kernels_ms", "end_to_end", "given_ops_step", "fused_step", "roofline")}))
