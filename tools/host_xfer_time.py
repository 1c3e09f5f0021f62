#!/usr/bin/env python3
"""Break-down of one host-pointer transportmatrix call at 1 degree: plan (uploads + count) and fetch (fill + downloads),
with output arrays fresh from np.empty (page faults inside the timed region, like a Julia caller's new SparseMatrixCSC)
and pre-touched.   gpurun -- python tools/host_xfer_time.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otmb_amd
import otmb_amd.api as api
from otmb_amd import capi, synthetic

g = synthetic.preset("access1deg", rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
idx = api.makeindices(gm.v3D)
phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
ctx = api.context(0)
lib = capi.lib()
N = int(idx["N"])
for reuse in (False, True):
    for touched in (False, True):
        for rep in range(3):
            keep = []
            a = api._tm_args(phi, g.mlotst, gm, idx, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, True, keep)
            ctx.set_reuse_grid(reuse)
            nnz = (C.c_int64 * 5)()
            t0 = time.perf_counter()
            ctx.check(lib.otmb_transportmatrix_plan(ctx.handle, C.byref(a), C.byref(nnz)))
            t1 = time.perf_counter()
            mk = np.zeros if touched else np.empty
            colptr = [mk(N + 1, dtype=np.int64) for _ in range(5)]
            rowval = [mk(int(nnz[m]), dtype=np.int64) for m in range(5)]
            nzval = [mk(int(nnz[m]), dtype=np.float64) for m in range(5)]
            cp = capi.ptr_array(5, [x.ctypes.data for x in colptr])
            rv = capi.ptr_array(5, [x.ctypes.data for x in rowval])
            nz = capi.ptr_array(5, [x.ctypes.data for x in nzval])
            final = (C.c_int64 * 5)()
            t2 = time.perf_counter()
            ctx.check(lib.otmb_transportmatrix_fetch(ctx.handle, C.byref(cp), C.byref(rv), C.byref(nz), C.byref(final)))
            t3 = time.perf_counter()
        up = sum(x.nbytes for x in keep)
        down = sum(x.nbytes for x in colptr + rowval + nzval)
        print(f"reuse_grid={reuse} outputs pre-touched={touched}: plan {1e3 * (t1 - t0):6.1f} ms ({up / 1e6:.0f} MB host arrays) | alloc {1e3 * (t2 - t1):6.1f} ms | "
              f"fetch {1e3 * (t3 - t2):6.1f} ms ({down / 1e6:.0f} MB, {down / (t3 - t2) / 1e9:.1f} GB/s)", flush=True)
