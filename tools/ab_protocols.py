#!/usr/bin/env python3
"""A/B of the two transportmatrix protocols (one process, interleaved rounds): per-kernel HIP-event times."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler

wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
nx, ny, nz, lf = synthetic.PRESETS[wl]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
for onepass in (True, False):
    asm.step(umo, vmo, 1e20, onepass=onepass)
asm.ctx.timing_enable(True)
for rnd in range(3):
    for onepass in (True, False):
        for _ in range(10):
            asm.step(umo, vmo, 1e20, onepass=onepass)
        kt = asm.ctx.timing_collect()
        print("onepass" if onepass else "twophase", {k: round(v[0] / v[1], 4) for k, v in kt.items()})
