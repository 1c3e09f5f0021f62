#!/usr/bin/env python3
"""Does the fill pass's time at 1 degree depend on which allocations its INPUT arrays live in?  One process, one assembler; the grid tensors (v3D, thkcello,
Lwet3D, Lwet, rho, the ten 2-D arrays, zt, wet mask / flags) are cloned NCOPIES times (a spacer allocation of random size before each copy), the assembler is pointed at
each copy in turn and facefluxes / counting pass / fill pass are timed (HIP events of the library), ROUNDS times.
    python tools/placement_inputs.py [NCOPIES] [ROUNDS]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler

ncopies = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
g = synthetic.make_grid(360, 300, 50, seed=20260501, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=True)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
fill = g.umo.properties["_FillValue"]
NAMES = ["v3d", "thk", "lwet3d", "lwet", "rho", "area", "zt", "mlotst", "wet3d", "wetflags"]
LISTS = ["edge", "dist"]
copies = [{n: getattr(asm, n) for n in NAMES} | {n: list(getattr(asm, n)) for n in LISTS}]
rng = np.random.default_rng(11)
for c in range(1, ncopies):
    spacer = torch.empty(int(rng.integers(3, 300)) * (1 << 20), dtype=torch.uint8, device=dev)
    copies.append({n: getattr(asm, n).clone() for n in NAMES} | {n: [t.clone() for t in getattr(asm, n)] for n in LISTS})
    del spacer
res = {k: [] for k in range(ncopies)}
for r in range(rounds):
    for k, cp in enumerate(copies):
        for n, v in cp.items():
            setattr(asm, n, v)
        asm._wetflags_version = asm.wet3d._version
        for _ in range(3):
            asm.step_async(umo, vmo, fill)
        asm.finish()
        asm.ctx.timing_enable(True)
        for _ in range(10):
            asm.step_async(umo, vmo, fill)
        asm.finish()
        asm.ctx.synchronize()
        t = asm.ctx.timing_collect()
        asm.ctx.timing_enable(False)
        res[k].append({kk: round(v[0] / v[1], 5) for kk, v in t.items() if kk in ("tm_kernel<fill>", "facefluxes_kernel", "tm_count_kernel")})
print(json.dumps({"fill_ff_count_ms_by_input_copy": res}))
