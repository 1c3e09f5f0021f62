#!/usr/bin/env python3
"""Where does the placement effect come from?  N assemblers on one 1-degree grid in one process: every array's device address and the
assembler's fill time, as JSON lines (analysed offline).   python tools/placement_dump.py [--n 12] [--workload access1deg]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=12); ap.add_argument("--workload", default="access1deg")
a = ap.parse_args()
import numpy as np, torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
from otmb_amd.capi import MATS
dev = torch.device("cuda", 0)
nx, ny, nz, lf = synthetic.PRESETS[a.workload]
g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev); vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
asms = []
for q in range(a.n):
    x = DeviceAssembler(0); x.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    for _ in range(3): x.step_async(umo, vmo, 1e20)
    x.finish(); asms.append(x)
times = [[] for _ in asms]
for rnd in range(3):
    for q, x in enumerate(asms):
        x.ctx.timing_enable(True)
        for _ in range(20): x.step_async(umo, vmo, 1e20)
        x.finish(); kt = x.ctx.timing_collect(); x.ctx.timing_enable(False)
        times[q].append({k: v[0] / v[1] for k, v in kt.items()})
for q, x in enumerate(asms):
    addr = {"v3d": x.v3d.data_ptr(), "thk": x.thk.data_ptr(), "rho": x.rho.data_ptr(), "lwet3d": x.lwet3d.data_ptr(), "lwet": x.lwet.data_ptr(),
            "area": x.area.data_ptr(), "mlotst": x.mlotst.data_ptr(), "push_mask": x.push_mask.data_ptr(), "wetflags": x.wetflags.data_ptr()}
    for k, t in enumerate(x.phi): addr[f"phi{k}"] = t.data_ptr()
    for k, t in enumerate(x.edge): addr[f"edge{k}"] = t.data_ptr()
    for k, t in enumerate(x.dist): addr[f"dist{k}"] = t.data_ptr()
    for m in MATS:
        for k, nm in enumerate(("colptr", "rowval", "nzval")): addr[f"{m}_{nm}"] = x.out[m][k].data_ptr()
    fill = float(np.median([t["tm_kernel<fill>"] for t in times[q]]))
    ff = float(np.median([t["facefluxes_kernel"] for t in times[q]]))
    print(json.dumps({"asm": q, "fill_ms": round(fill, 4), "facefluxes_ms": round(ff, 4), "addr": addr}), flush=True)
