#!/usr/bin/env python3
"""lump_and_spray on the resident 1 degree T, a few calls, for a rocprofv3 --kernel-trace --stats run (which of its kernels takes the time?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
asm.step(umo, vmo, 1e20)
for _ in range(6):
    asm.lump_and_spray(None, 2, 2, 1)
torch.cuda.synchronize()
