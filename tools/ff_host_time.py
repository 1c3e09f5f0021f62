#!/usr/bin/env python3
"""The host-pointer facefluxes on the 1 degree grid: single context against S depth slabs of the same GPU taking the link in turn
(otmb_mgpu_facefluxes).   gpurun -- python tools/ff_host_time.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import otmb_amd
import otmb_amd.api as api
from otmb_amd import synthetic

nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
idx = api.makeindices(gm.v3D)
for slabs in (0, 2, 3, 4, 6, 8, 0, 4):
    ts = []
    for rep in range(9):
        t0 = time.perf_counter()
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=([0] * slabs if slabs else None))
        ts.append(time.perf_counter() - t0)
        del phi
    print(json.dumps({"slabs": slabs, "facefluxes_ms": round(1e3 * float(np.median(ts[3:])), 2), "min_ms": round(1e3 * min(ts[3:]), 2)}), flush=True)
