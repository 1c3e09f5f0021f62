#!/usr/bin/env python3
"""Is it WHERE an assembler's arrays lie or HOW MANY allocations came before?  Assemblers created after dummy blocks of various sizes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
dev = torch.device("cuda", 0)
nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev); vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
def mk():
    x = DeviceAssembler(0); x.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    for _ in range(3): x.step_async(umo, vmo, 1e20)
    x.finish(); return x
def t(x):
    r = []
    for _ in range(3):
        x.ctx.timing_enable(True)
        for _ in range(20): x.step_async(umo, vmo, 1e20)
        x.finish(); kt = x.ctx.timing_collect(); x.ctx.timing_enable(False)
        r.append(kt["tm_kernel<fill>"][0] / kt["tm_kernel<fill>"][1])
    return round(float(np.median(r)), 4)
keep = []
for label, dummy_gb in (("fresh", 0), ("after 4 GB", 4), ("after 40 GB more", 40), ("after 100 GB more", 100), ("after 60 GB more", 60)):
    if dummy_gb:
        keep.append(torch.empty(int(dummy_gb * 1e9), dtype=torch.uint8, device=dev))
    x = mk(); keep.append(x)
    print(json.dumps({"assembler": label, "fill_ms": t(x), "v3d_addr": hex(x.v3d.data_ptr()), "T_rowval_addr": hex(x.out["T"][1].data_ptr())}), flush=True)
print(json.dumps({"again": [t(x) for x in keep if isinstance(x, DeviceAssembler)]}))
# free the dummies and build one more: does it land high again, and is it fast?
keep = [x for x in keep if isinstance(x, DeviceAssembler)]
import gc; gc.collect(); torch.cuda.empty_cache()
x = mk()
print(json.dumps({"assembler": "after freeing the dummies", "fill_ms": t(x), "v3d_addr": hex(x.v3d.data_ptr())}))
