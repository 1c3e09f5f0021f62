#!/usr/bin/env python3
"""End-to-end time of the HOST-pointer API (what Julia's ccall path costs, PCIe included) on the 1° grid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otmb_amd
from otmb_amd import synthetic, api
nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
idx = api.makeindices(gm.v3D)
for rep in range(3):
    t0 = time.perf_counter(); idx = api.makeindices(gm.v3D); t1 = time.perf_counter()
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx); t2 = time.perf_counter()
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho); t3 = time.perf_counter()
    tt = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, operators=False); t4 = time.perf_counter()
    print(f"makeindices {1e3*(t1-t0):.1f} ms, facefluxes {1e3*(t2-t1):.1f} ms, transportmatrix {1e3*(t3-t2):.1f} ms, "
          f"transportmatrix(operators=false) {1e3*(t4-t3):.1f} ms  (N={idx.N}, nnz(T)={tm.T.nnz})")
