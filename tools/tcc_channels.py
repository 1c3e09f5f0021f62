#!/usr/bin/env python3
"""Per-instance (L2 channel x XCD) values of raw TCC counters from a rocprofv3 `--output-format json` run: how evenly do the
128 L2 channels share a kernel's requests?    Usage: tcc_channels.py <dir> <kernel substring> > summary.json
The JSON layout differs between rocprofv3 versions, so this walks it defensively and keeps what it cannot interpret."""
import collections, glob, json, sys

root, want = sys.argv[1], sys.argv[2]
out = {"files": [], "kernels": {}}
for f in glob.glob(root + "/**/*results.json", recursive=True):
    d = json.load(open(f))
    tool = d.get("rocprofiler-sdk-tool", d)
    tool = tool[0] if isinstance(tool, list) else tool
    out["files"].append(f)
    # kernel symbols: id -> name
    names = {}
    for ks in tool.get("kernel_symbols", []):
        names[ks.get("kernel_id")] = ks.get("formatted_kernel_name") or ks.get("kernel_name")
    # counter metadata: id -> name (+ dimensions when present)
    cmeta = {}
    for c in tool.get("counters", []):
        cmeta[c.get("id", {}).get("handle", c.get("id"))] = {"name": c.get("name"), "dims": c.get("dimension_ids", c.get("dimensions"))}
    recs = tool.get("callback_records", {}).get("counter_collection", []) or tool.get("buffer_records", {}).get("counter_collection", [])
    if recs:
        out.setdefault("example_record_keys", sorted(recs[0].keys()))
    per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter name -> [per-dispatch lists of instance values]
    for r in recs:
        di = r.get("dispatch_data", {}).get("dispatch_info", {})
        kname = names.get(di.get("kernel_id"), str(di.get("kernel_id")))
        if want not in (kname or ""):
            continue
        inst = collections.defaultdict(list)
        for x in r.get("records", []):
            cid = x.get("counter_id", {})
            cid = cid.get("handle", cid) if isinstance(cid, dict) else cid
            nm = cmeta.get(cid, {}).get("name")
            if nm is None:  # instance records carry the dimension position in the id's upper bits on some versions
                nm = cmeta.get(cid & 0xFFFF, {}).get("name", str(cid)) if isinstance(cid, int) else str(cid)
            inst[nm].append(x.get("value"))
        for nm, vals in inst.items():
            per[kname][nm].append(vals)
    if "--per-dispatch" in sys.argv:  # one row per dispatch and counter: the instances' sum, min / max SHARE, busiest over mean
        for kname, cs in per.items():
            k = out.setdefault("per_dispatch", {}).setdefault(kname[:80], {})
            for nm, disp in cs.items():
                rows = []
                for v in disp:
                    tot = float(sum(v)) or 1.0
                    rows.append({"instances": len(v), "sum": tot, "min_share": min(v) / tot, "max_share": max(v) / tot, "max_over_mean": max(v) * len(v) / tot,
                                 "stdev_over_mean": (sum((x - tot / len(v)) ** 2 for x in v) / len(v)) ** 0.5 / (tot / len(v))})
                k[nm] = rows
    for kname, cs in per.items():
        k = out["kernels"].setdefault(kname[:80], {})
        for nm, disp in cs.items():
            n = max(len(v) for v in disp)
            full = [v for v in disp if len(v) == n]
            mean = [sum(v[i] for v in full) / len(full) for i in range(n)]
            tot = sum(mean)
            k[nm] = {"instances": n, "dispatches": len(full), "sum": tot, "min": min(mean), "max": max(mean),
                     "max_over_mean": (max(mean) * n / tot) if tot else None, "per_instance_mean": [round(x, 1) for x in mean]}
    out["counter_meta_sample"] = dict(list(cmeta.items())[:3])
print(json.dumps(out, indent=1))
