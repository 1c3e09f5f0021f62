#!/usr/bin/env python3
"""The host-pointer transportmatrix on the 1 degree grid: the two-phase call (plan -> allocate -> fetch: every upload before the count, every
download after it) against the pipelined one-phase build on S depth slabs of the same GPU (api.transportmatrix(..., slabs=S):
otmb_mgpu_transportmatrix_onepass -- a slab uploads while the one above it copies its columns home).  Without and with the reuse promises.
gpurun -- python tools/onepass_time.py"""
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
phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)


def timed(slabs, reuse, reps=5):
    ts = []
    for rep in range(reps + 2):
        t0 = time.perf_counter()
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, slabs=slabs, reuse_grid=reuse, reuse_fluxes=False)
        t1 = time.perf_counter()
        if rep > 1:
            ts.append(t1 - t0)
        del tm
    return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3


for slabs in (None, 1, 2, 3, 4, 6, 8, 12):
    for reuse in (False, True):
        med, best = timed(slabs, reuse)
        print(json.dumps({"slabs": slabs or 0, "reuse_grid": reuse, "transportmatrix_ms": round(med, 2), "min_ms": round(best, 2)}), flush=True)
