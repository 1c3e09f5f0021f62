#!/usr/bin/env python3
"""Interleaved in-process A/B of library knobs that a context reads from the environment when it is created (OTMB_FF_XCD, OTMB_COUNT_ORDER,
OTMB_PF_DIST, OTMB_MARCH_ROWS ...): several assemblers per variant (placement moves the fill pass by +-5 %, profiles/r03/README.md section 6),
created interleaved, timed round-robin with HIP events on the launch stream.
   python tools/env_ab.py --variants "base:OTMB_PF_DIST=0;pf96:OTMB_PF_DIST=96" [--lib name] [--workload access1deg] [--reps 3] [--rounds 4] [--steps 20]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--variants", required=True)
ap.add_argument("--lib", default="default")
ap.add_argument("--workload", default="access1deg")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
import numpy as np, torch
import otmb_amd
from otmb_amd import capi, synthetic, synthetic_device
from otmb_amd.device import DeviceAssembler
if a.lib != "default":
    capi.use_library(os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "lib", f"libotmb_hip_{a.lib}.so"), lenient=True)
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
variants = []
for v in a.variants.split(";"):
    name, _, kv = v.partition(":")
    variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
allkeys = {k for _, e in variants for k in e}
asms = []
for rep in range(a.reps):
    vs = variants[::-1] if rep % 2 else variants
    for name, env in vs:
        for k in allkeys:
            os.environ.pop(k, None)
        os.environ.update(env)
        asms.append((name, mk()))
for k in allkeys:
    os.environ.pop(k, None)
res = {}
for rnd in range(a.rounds):
    order = asms[::-1] if rnd % 2 else asms
    for name, x in order:
        for _ in range(3): x.step_async(umo, vmo, fill)
        x.finish(); x.ctx.timing_enable(True)
        for _ in range(a.steps): x.step_async(umo, vmo, fill)
        x.finish(); kt = x.ctx.timing_collect(); x.ctx.timing_enable(False)
        for kk, vv in kt.items():
            res.setdefault(name, {}).setdefault(kk, {}).setdefault(id(x), []).append(vv[0] / vv[1])
alg = asms[0][1].algorithmic_bytes()
for name, _ in variants:
    rec = {"variant": name, "workload": a.workload}
    tot = 0.0
    for kk, per in res[name].items():
        med = [float(np.median(v)) for v in per.values()]  # per assembler
        rec[kk] = {"mean_ms": round(float(np.mean(med)), 5), "per_assembler": [round(q, 5) for q in med]}
        tot += float(np.mean(med))
    rec["sum_ms"] = round(tot, 5)
    if "tm_kernel<fill>" in rec:
        rec["fill_frac_of_8TBs"] = round(alg / rec["tm_kernel<fill>"]["mean_ms"] / 1e9 / 8, 4)
    print(json.dumps(rec), flush=True)
