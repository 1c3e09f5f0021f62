#!/usr/bin/env python3
"""One time-slice loop of the host-pointer API on the 1 degree grid, for A/B runs in fresh processes on ONE box (tools/host_ab.sh):
facefluxesfrommasstransport + transportmatrix with explicit protocols (slabs = 4 pipelined, slabs = 0 two-phase) and with TκH / TκVdeep passed
back; prints medians (ms) of the steady slices.  The library's switches come from the environment (OTMB_XFER_NARROW, OTMB_FF_SHIFT_ON_HOST ...)."""
import json
import os
import sys
import time

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
H = api.buildTκH(gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH)
D = api.buildTκVdeep(mlotst=g.mlotst, gridmetrics=gm, indices=idx, κVdeep=g.kappaVdeep)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("OTMB_")}}
legs = (("pipelined", dict(slabs=4)), ("two_phase", dict(slabs=0)), ("given_pipelined", dict(slabs=4, TκH=H, TκVdeep=D, reuse_grid=True)),
        ("given_two_phase", dict(slabs=0, TκH=H, TκVdeep=D, reuse_grid=True)))
for name, kw in legs:
    ff, tm = [], []
    for rep in range(9):
        t0 = time.perf_counter()
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
        t1 = time.perf_counter()
        r = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH, κVML=g.kappaVML, κVdeep=g.kappaVdeep, **kw)
        t2 = time.perf_counter()
        if rep >= 3:
            ff.append(t1 - t0)
            tm.append(t2 - t1)
        del r, phi
    out[name] = {"facefluxes_ms": round(1e3 * float(np.median(ff)), 2), "transportmatrix_ms": round(1e3 * float(np.median(tm)), 2),
                 "transportmatrix_min_ms": round(1e3 * float(np.min(tm)), 2)}
print(json.dumps(out), flush=True)
