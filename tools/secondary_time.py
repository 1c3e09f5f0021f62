#!/usr/bin/env python3
"""The supporting kernels of the path on the 1 degree grid, device resident: time per call (torch events on the
launch stream around the C-ABI call, median of REPS after warm-up), the ALGORITHMIC bytes of the call (inputs read once + outputs written once) and what fraction of the
8 TB/s HBM peak that is.  Multi-kernel calls (sparse(): pack + radix sort + segmented sum; lump_and_spray: sweep + components + scans / sort) are reported per call.
    python tools/secondary_time.py [access1deg|quarterdeg]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import otmb_amd
from otmb_amd import capi, synthetic
from otmb_amd.capi import MATS
from otmb_amd.device import DeviceAssembler

wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
REPS = 7
dev = torch.device("cuda", 0)
nx, ny, nz, lf = synthetic.PRESETS[wl]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
fill = g.umo.properties["_FillValue"]
G, N, P = asm.G, asm.N, nx * ny
lib, ctx = asm.lib, asm.ctx
rows = []


def timed(name, nbytes, call, note=""):
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    rows.append({"call": name, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1), "TBps": round(nbytes / (ms * 1e-3) / 1e12, 3),
                 "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 8e12, 3), "note": note})
    print(json.dumps(rows[-1]), flush=True)


# a3 makeindices: v3D in; wet byte, Lwet3D, Lwet out
_n = C.c_int64(0)
_lw3, _lw, _w3 = torch.empty_like(asm.lwet3d), torch.empty_like(asm.lwet), torch.empty_like(asm.wet3d)
timed("otmb_makeindices_dev", 8 * G + G + 8 * G + 8 * N, lambda: ctx.check(lib.otmb_makeindices_dev(ctx.handle, asm.v3d.data_ptr(), nx, ny, nz, _lw3.data_ptr(), _lw.data_ptr(), _w3.data_ptr(), C.byref(_n))),
      "two kernels (count, write) + a host round trip for N")
# wet flags (grid constant of nofluxboundaries!): wet byte in, flag byte out
timed("otmb_wetflags_dev", 2 * G, lambda: ctx.check(lib.otmb_wetflags_dev(ctx.handle, asm.wet3d.data_ptr(), nx, ny, nz, asm.topology, asm.wetflags.data_ptr())))
phi = asm.facefluxes(umo, vmo, fill)
# push mask from existing fluxes: six flux arrays + Lwet3D in, 2 bytes out
ptrs = capi.ptr_array(6, [p.data_ptr() for p in phi])
mask2 = torch.empty_like(asm.push_mask)
timed("otmb_push_mask_dev", 48 * G + 8 * G + 2 * G, lambda: ctx.check(lib.otmb_push_mask_dev(ctx.handle, C.byref(ptrs), asm.lwet3d.data_ptr(), 0, G, mask2.data_ptr())))
# f1 velocity2fluxes / fluxes2velocity: u, v, rho, thk in (+ two 2-D edge arrays); two arrays out
u = torch.where(asm.wet3d != 0, torch.randn(G, dtype=torch.float64, device=dev) * 0.1, torch.full((G,), 1e20, dtype=torch.float64, device=dev))
v = torch.where(asm.wet3d != 0, torch.randn(G, dtype=torch.float64, device=dev) * 0.1, torch.full((G,), 1e20, dtype=torch.float64, device=dev))
fi, fj = torch.empty_like(u), torch.empty_like(u)
e_east, e_north = asm.edge[capi.HDIRS.index("east")], asm.edge[capi.HDIRS.index("north")]
for fn in ("otmb_velocity2fluxes_dev", "otmb_fluxes2velocity_dev"):
    timed(fn, 8 * G * 6 + 16 * P, lambda fn=fn: ctx.check(getattr(lib, fn)(ctx.handle, u.data_ptr(), v.data_ptr(), 0, asm.rho.data_ptr(), 0.0, asm.thk.data_ptr(), e_east.data_ptr(),
                                                                        e_north.data_ptr(), nx, ny, nz, asm.topology, fi.data_ptr(), fj.data_ptr())))
# f4 B-grid -> C-grid interpolation: two arrays in, two out
u2, v2 = torch.empty_like(u), torch.empty_like(u)
timed("otmb_bgrid_to_cgrid_dev", 8 * G * 4, lambda: ctx.check(lib.otmb_bgrid_to_cgrid_dev(ctx.handle, u.data_ptr(), v.data_ptr(), 0, 1e20, nx, ny, nz, u2.data_ptr(), v2.data_ptr())))
# f2 makegridmetrics on the device: the library's own HIP events around its kernels (the call also uploads the raw CMIP arrays from the host)
asm2 = DeviceAssembler(0)
asm2.ctx.timing_enable(True)
for _ in range(3):
    asm2.set_grid_from_raw(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices,
                           mlotst=g.mlotst, rho=g.rho)
asm2.ctx.synchronize()
tg = asm2.ctx.timing_collect()
asm2.ctx.timing_enable(False)
for kname, (ms_sum, cnt) in tg.items():
    if "gridmetrics" in kname:
        nb = 8 * G * 4 + 8 * P * 30  # volcello in, v3D / thkcello / Z3D out; the 2-D fields (vertices, edges, distances): nominal
        rows.append({"call": f"otmb_makegridmetrics_dev: {kname}", "ms": round(ms_sum / cnt, 4), "algorithmic_MB": round(nb / 1e6, 1), "TBps": round(nb / (ms_sum / cnt * 1e-3) / 1e12, 3),
                     "frac_of_8TBps": round(nb / (ms_sum / cnt * 1e-3) / 8e12, 3), "note": "kernels only (HIP events of the library); 12 haversines per (i, j) + the 3-D arrays"})
        print(json.dumps(rows[-1]), flush=True)
del asm2
# a15 bolus_GM_velocity: rho, Z3D, wet byte in (+ two 2-D distances); u, v out
dn = gm.distance_to_neighbour_2D
flat = lambda a: torch.from_numpy(np.asfortranarray(a, dtype=np.float64).ravel(order="F")).to(dev)
z3d, de, dnn = flat(gm.Z3D), flat(dn["east"]), flat(dn["north"])
bu, bv = torch.empty_like(u), torch.empty_like(u)
timed("otmb_bolus_gm_velocity_dev", 8 * G * 4 + G + 16 * P, lambda: ctx.check(lib.otmb_bolus_gm_velocity_dev(ctx.handle, asm.rho.data_ptr(), z3d.data_ptr(), asm.wet3d.data_ptr(), de.data_ptr(),
                                                                                                     dnn.data_ptr(), nx, ny, nz, asm.topology, 600.0, 0.01, bu.data_ptr(), bv.data_ptr())),
      "two kernels; the two kappaGM*S arrays between them are written and read once more (not counted)")
# the fused transportmatrix, for the operators below
asm.transportmatrix(phi)
nnz = dict(zip(MATS, asm.nnz))
out = asm.out
# a11 general path: COO generators + sparse() per operator (entries: 24 B each written, then read; CSC written)
for m in MATS[1:]:
    I, J, V = asm.sparse_entries(m)
    E = I.numel()
    timed(f"sparse_entries({m})", 24 * E + 8 * G * 4, lambda m=m: asm.sparse_entries(m), f"{E} triplets (plan + fill: two passes over the grid)")
    timed(f"sparse({m})", 24 * E + 16 * nnz[m] + 8 * (N + 1), lambda I=I, J=J, V=V: asm.sparse(I, J, V, N, N), "pack keys + rocPRIM radix sort + segmented sum (plan + fill)")
# a12 sparse add (precomputed operators): two CSC in, one out
A = out["Tadv"]
B = out["TκH"]
cp, rv, nz_ = asm.spadd(A, B, N)
timed("spadd(Tadv, TκH)", 16 * (nnz["Tadv"] + nnz["TκH"] + rv.numel()) + 24 * (N + 1), lambda: asm.spadd(A, B, N), "plan + fill")
# f3 lump_and_spray on the resident T
timed("lump_and_spray(2x2x1)", 16 * nnz["T"] + 8 * (N + 1) + 8 * G + 40 * N, lambda: asm.lump_and_spray(None, 2, 2, 1), "sweep + components + scans / sort; bytes: T's pattern + volumes in, LUMP / SPRAY out (nominal)")
print(json.dumps({"workload": wl, "G": G, "N": N, "rows": rows}))
