#!/usr/bin/env python3
"""OTMB_ONEPASS_TRACE=1: the phase stamps of every slab of the pipelined host-pointer transportmatrix (1 degree grid), 6th call.
gpurun -- 'OTMB_ONEPASS_TRACE=1 python tools/onepass_trace.py [slabs]'"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import otmb_amd
import otmb_amd.api as api
from otmb_amd import synthetic

slabs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
idx = api.makeindices(gm.v3D)
phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
for rep in range(6):
    if rep == 5:
        sys.stderr.write(f"---- call {rep + 1}, {slabs} slabs\n")
    else:
        sys.stderr.flush()
    t0 = time.perf_counter()
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, slabs=slabs)
    dt = time.perf_counter() - t0
    del tm
sys.stderr.write(f"last call {dt * 1e3:.2f} ms\n")
