#!/usr/bin/env python3
"""Seam row or not: kernel times of the 1-degree workload on a tripolar and on a bipolar grid (same mask, same fluxes).
The cells of a tripolar grid's seam row take the generic column builder in the counting and fill passes.
   gpurun -- python tools/topology_scan.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler

nx, ny, nz, lf = synthetic.PRESETS[os.environ.get("WORKLOAD", "access1deg")]
for topology in ("tripolar", "bipolar"):
    g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array", topology=topology)
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                  lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    if True:
        a = DeviceAssembler(0)
        a.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
        for _ in range(3):
            a.step(umo, vmo, 1e20)
        a.ctx.timing_enable(True)
        for _ in range(20):
            a.step(umo, vmo, 1e20)
        kt = {k: round(v[0] / v[1], 4) for k, v in a.ctx.timing_collect().items()}
        print(topology, "N =", a.N, kt, flush=True)
        del a
