#!/usr/bin/env python3
"""Is there anything to choose at 1 degree?  DeviceAssembler.choose_placement with its size threshold off: the facefluxes / fill-pass times of
`candidates` flux sets / output sets inside ONE process.   gpurun -- python tools/placement_1deg.py [candidates]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=True)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
rec = asm.choose_placement(umo, vmo, 1e20, candidates=n, reps=20, min_output_bytes=0)
print(json.dumps({k: ([round(x, 4) for x in v] if isinstance(v, list) and v and isinstance(v[0], float) else v) for k, v in rec.items()}))
