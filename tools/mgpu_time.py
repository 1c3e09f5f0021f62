#!/usr/bin/env python3
"""What the slab orchestration behind the C ABI costs on ONE GPU (N contexts share the device and its PCIe link, so no speed-up can show here:
this is the overhead of threads, halo levels and the chain's hand-offs): facefluxesfrommasstransport + transportmatrix through the host-pointer
API on the 1 degree grid, single context against otmb_mgpu with 1, 2, 4, 8 slabs, without and with the reuse flags.   gpurun -- python tools/mgpu_time.py"""
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

def timed(devices, reuse, reps=4):
    ts = []
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
        t1 = time.perf_counter()
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices, reuse_grid=reuse, reuse_fluxes=reuse)
        t2 = time.perf_counter()
        if rep:
            ts.append((t1 - t0, t2 - t1))
        del tm, phi
    return float(np.median([a for a, _ in ts])) * 1e3, float(np.median([b for _, b in ts])) * 1e3

for devices in (None, [0], [0, 0], [0, 0, 0, 0], [0] * 8):
    for reuse in (False, True):
        ff, tm = timed(devices, reuse)
        print(json.dumps({"slabs": 0 if devices is None else len(devices), "reuse_flags": reuse, "facefluxes_ms": round(ff, 2), "transportmatrix_ms": round(tm, 2),
                          "wet_cells_per_s": round(int(idx.N) / ((ff + tm) * 1e-3))}), flush=True)
