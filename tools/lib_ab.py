#!/usr/bin/env python3
"""Interleaved A/B of prebuilt library variants x tile orders on one device-generated grid (one assembler per variant, kept):
   python tools/lib_ab.py --libs default,nt --rows 0,8 [--workload quarterdeg] [--rounds 4] [--steps 10]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--libs", default="default"); ap.add_argument("--rows", default="0"); ap.add_argument("--workload", default="quarterdeg")
ap.add_argument("--rounds", type=int, default=4); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--stagger", default="0", help="OTMB_STAGGER values (bytes): the k-th array of an assembler starts k*stagger bytes into its allocation")
ap.add_argument("--bipolar", action="store_true", help="also time every assembler with the topology flag forced to bipolar (no seam row: what the generic column path costs)")
ap.add_argument("--only-bipolar", action="store_true", help="time ONLY with the topology flag forced to bipolar (variants that cannot do the seam row)")
ap.add_argument("--reps", type=int, default=1, help="assemblers per library (the placement of the arrays in HBM moves the fill pass by +-5 %)")
a = ap.parse_args()
import numpy as np, torch
import otmb_amd
from otmb_amd import capi, synthetic, synthetic_device
from otmb_amd.device import DeviceAssembler
dev = torch.device("cuda", 0)
if a.workload in ("quarterdeg", "tenthdeg"):
    dg = synthetic_device.make_device_grid(a.workload, dev, seed=20260501, rho="array"); umo, vmo, fill = dg.umo, dg.vmo, dg.fill
    mk = lambda: synthetic_device.assembler_for(dg, 0)
else:
    nx, ny, nz, lf = synthetic.PRESETS[a.workload]
    g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev); vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev); fill = 1e20
    def mk():
        x = DeviceAssembler(0); x.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep); return x
asms = []
for rep in range(a.reps):
    names = a.libs.split(",")
    if rep % 2: names.reverse()
    for name in names:
        for stg in a.stagger.split(","):
            capi.use_library(os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "lib", "libotmb_hip.so" if name == "default" else f"libotmb_hip_{name}.so"), lenient=True)
            os.environ["OTMB_STAGGER"] = stg
            x = mk(); x.ctx.set_formulation(0); asms.append((name + ("" if stg == "0" else "+stagger" + stg), rep, x))
rows = [int(r) for r in a.rows.split(",")]
res = {}
allk = {}
for rnd in range(a.rounds):
    order = [(n, rep, x, r) for n, rep, x in asms for r in rows]
    if rnd % 2: order.reverse()
    if a.only_bipolar:
        order = [(n + "+bipolar", rep, x, r, 0) for n, rep, x, r in order]
    elif a.bipolar:
        order = [(n + tag, rep, x, r, topo) for n, rep, x, r in order for tag, topo in (("", None), ("+bipolar", 0))]
    else:
        order = [(n, rep, x, r, None) for n, rep, x, r in order]
    for n, rep, x, r, topo in order:
        if not hasattr(x, "_topo0"):
            x._topo0 = x.topology
        x.topology = x._topo0 if topo is None else topo
        if a.bipolar or a.only_bipolar:
            x.makeindices()  # (the folded wet flags depend on the topology)
        x.ctx.set_tile_order(r)
        for _ in range(3): x.step_async(umo, vmo, fill)
        x.finish(); x.ctx.timing_enable(True)
        for _ in range(a.steps): x.step_async(umo, vmo, fill)
        x.finish(); kt = x.ctx.timing_collect(); x.ctx.timing_enable(False)
        res.setdefault((n, r), {}).setdefault(rep, []).append(kt["tm_kernel<fill>"][0] / kt["tm_kernel<fill>"][1])
        for kk, vv in kt.items():
            allk.setdefault((n, r), {}).setdefault(kk, []).append(vv[0] / vv[1])
alg = asms[0][2].algorithmic_bytes()
for (n, r), d in res.items():
    per = [float(np.median(v)) for v in d.values()]  # per assembler
    m = float(np.mean(per))
    print(json.dumps({"lib": n, "rows": r, "fill_ms_mean_over_assemblers": round(m, 4), "per_assembler": [round(q, 4) for q in per],
                      "frac_of_8TBs": round(alg / m / 1e9 / 8, 3),
                      "kernels_ms": {kk: round(float(np.mean(vv)), 4) for kk, vv in allk[(n, r)].items()},
                      "sum_ms": round(float(sum(np.mean(vv) for vv in allk[(n, r)].values())), 4)}), flush=True)
