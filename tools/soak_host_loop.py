#!/usr/bin/env python3
"""A long time loop through the host-pointer API (facefluxes + the default transportmatrix, 1 degree grid): host RSS, pinned pool and device memory
must stay flat after the first slices.   gpurun -- python tools/soak_host_loop.py [slices]"""
import gc, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import psutil
import torch
import otmb_amd
import otmb_amd.api as api
from otmb_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
idx = api.makeindices(gm.v3D)
proc = psutil.Process()
ref_nnz, t0 = None, time.perf_counter()
for k in range(n):
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho)
    nnz = [tm[m].nnz for m in ("T", "Tadv", "TκH", "TκVML", "TκVdeep")]
    chk = float(tm["T"].nzval[:: 4097].sum())
    if ref_nnz is None:
        ref_nnz, ref_chk = nnz, chk
    assert nnz == ref_nnz and chk == ref_chk, (k, nnz, chk)
    del tm, phi
    if k in (9, 19, 49, 99, 199, n - 1):
        gc.collect()
        free_b, total_b = torch.cuda.mem_get_info(0)
        print(json.dumps({"slice": k + 1, "host_rss_gb": round(proc.memory_info().rss / 2 ** 30, 3), "device_used_gb": round((total_b - free_b) / 2 ** 30, 3),
                          "protocol": "pipelined" if api.Trial.of(0, int(idx["N"])).now else "two-phase", "s_per_slice": round((time.perf_counter() - t0) / (k + 1), 4)}), flush=True)
