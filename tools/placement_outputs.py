#!/usr/bin/env python3
"""How much does the fill pass's time depend on WHICH allocation its output arrays live in, inside one process?  One assembler, NSETS output sets
(new_output_set(); a spacer allocation of a different size is made and freed before each, so that the sets do not simply reuse one another's blocks), the fill
pass timed on every set in turn, ROUNDS times (HIP events of the library, on the launch stream).  Reproducible per set => the choice can be made once at set-up.
    python tools/placement_outputs.py [access1deg|quarterdeg] [NSETS] [ROUNDS]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import otmb_amd
from otmb_amd import synthetic, synthetic_device
from otmb_amd.device import DeviceAssembler

wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
nsets = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
if wl == "quarterdeg":
    dg = synthetic_device.make_device_grid(wl, dev, seed=20260501, rho="array")
    asm = synthetic_device.assembler_for(dg, 0)
    umo, vmo, fill = dg.umo, dg.vmo, dg.fill
else:
    g = synthetic.make_grid(360, 300, 50, seed=20260501, rho="array")
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=True)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
    fill = g.umo.properties["_FillValue"]
phi = asm.facefluxes(umo, vmo, fill)
asm.transportmatrix_onepass(phi)  # the assembler's own set: index 0
sets = [asm.out]
rng = np.random.default_rng(7)
for s in range(1, nsets):
    spacer = torch.empty(int(rng.integers(3, 400)) * (1 << 20) + 4096 * int(rng.integers(0, 255)), dtype=torch.uint8, device=dev)
    sets.append(asm.new_output_set())
    del spacer
res = {k: [] for k in range(nsets)}
ffs = []
for r in range(rounds):
    for k, out in enumerate(sets):
        for _ in range(2):
            asm.transportmatrix_onepass(phi, out=out)
        asm.ctx.synchronize()
        asm.ctx.timing_enable(True)
        for _ in range(8):
            asm.transportmatrix_onepass(phi, sync=False, out=out)
        asm.result()
        asm.ctx.synchronize()
        t = asm.ctx.timing_collect()
        asm.ctx.timing_enable(False)
        res[k].append(round(t["tm_kernel<fill>"][0] / t["tm_kernel<fill>"][1], 5))
MATS = ("T", "Tadv", "TkH", "TkVML", "TkVdeep")
addrs = [{m: [hex(s[m][q].data_ptr()) for q in range(3)] for m in s} for s in sets]
inputs = {"phi": [hex(p.data_ptr()) for p in asm.phi], "v3d": hex(asm.v3d.data_ptr()), "thk": hex(asm.thk.data_ptr()), "lwet3d": hex(asm.lwet3d.data_ptr()),
          "lwet": hex(asm.lwet.data_ptr()), "rho": hex(asm.rho.data_ptr()) if asm.rho is not None else None}
print(json.dumps({"workload": wl, "fill_ms_by_output_set": res, "addr": addrs, "inputs": inputs, "nnz": asm.nnz, "N": asm.N}))
