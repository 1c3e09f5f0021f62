#!/usr/bin/env python3
"""bench.py's end_to_end loop, every repetition printed: facefluxes + transportmatrix through the host-pointer API on the 1 degree grid,
default (pipelined one-phase build) against slabs=0 (two-phase).   gpurun -- python tools/onepass_loop.py"""
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
for name, kw in (("default", {}), ("two_phase", {"slabs": 0}), ("default", {}), ("two_phase", {"slabs": 0})):
    ts = []
    for rep in range(8):
        t0 = time.perf_counter()
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
        t1 = time.perf_counter()
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH, κVML=g.kappaVML, κVdeep=g.kappaVdeep, **kw)
        t2 = time.perf_counter()
        ts.append((round(1e3 * (t1 - t0), 2), round(1e3 * (t2 - t1), 2), round(1e3 * (api.last_call_seconds["plan"] + api.last_call_seconds["fetch"]), 2)))
        del tm, phi
    print(name, ts, flush=True)
