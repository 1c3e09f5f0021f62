#!/usr/bin/env python3
"""Largest configuration on ONE GPU with the 3-D inputs generated on the device (the 0.1 degree grid has 729 M cells:
5.8 GB per Float64 array, too much to build on the host and copy):
   python tools/large_run_device.py tenthdeg|quarterdeg [steps]
2-D geometry comes from the host generator (synthetic.py) and the host makegridmetrics; thickness, volume, density
and the mass transports are the same formulas evaluated with torch on the GPU (own random stream, so the values differ
from synthetic.make_grid; the checks are size-independent properties, not oracle comparisons).  Outputs are sized
exactly by the two-phase protocol (plan -> fill).  Checks run over column chunks to bound their temporaries."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import otmb_amd
from otmb_amd import synthetic
from otmb_amd._nt import Cube
from otmb_amd.capi import HDIRS, MATS
from otmb_amd.device import DeviceAssembler

wl = sys.argv[1] if len(sys.argv) > 1 else "tenthdeg"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nx, ny, nz, lf = synthetic.PRESETS[wl]
P, G = nx * ny, nx * ny * nz
dev = torch.device("cuda", 0)
FILL = synthetic.FILL
t0 = time.time()

# ---- 2-D part on the host (as synthetic.make_grid) ----
rng = np.random.default_rng(20260501)
zt, dz = synthetic.levels(nz)
zbot = np.cumsum(dz)
lonv, latv = synthetic.vertices(nx, ny, "tripolar")
lon, lat = np.asfortranarray(lonv.mean(axis=0)), np.asfortranarray(latv.mean(axis=0))
lat_s, lat_n = np.deg2rad(latv[0]), np.deg2rad(np.maximum(latv[3], latv[2]))
area = np.asfortranarray(np.maximum(synthetic.R**2 * np.deg2rad(360.0 / nx) * np.abs(np.sin(lat_n) - np.sin(lat_s)), 1.0e6))
scale = max(3, min(nx, ny) // 12)
f = synthetic._smooth_field(rng, nx, ny, scale)
land = f < np.quantile(f, lf)
mid = nx // 2
land[max(0, mid - 3): mid + 3, -3:] = False
land[:2, -2:] = False
land[-2:, -2:] = False
depth = zbot[-1] * (0.05 + 0.95 * synthetic._smooth_field(rng, nx, ny, max(2, scale // 2)))
depth = np.where(rng.random((nx, ny)) < 0.02, 0.6 * dz[0], depth)
depth = np.where(land, 0.0, depth)
mlotst = np.exp(rng.uniform(np.log(10.0), np.log(1000.0), (nx, ny)))
# geometry: makegridmetrics' 2-D outputs do not depend on volcello, so a one-level volume is enough here
vol1 = np.asfortranarray((np.clip(depth, 0.0, dz[0]) * area)[:, :, None])
gm = otmb_amd.makegridmetrics(areacello=Cube(area, _FillValue=FILL), volcello=Cube(vol1, _FillValue=FILL), lon=lon, lat=lat, lev=zt[:1],
                              lon_vertices=lonv, lat_vertices=latv)
print(f"[{wl}] host 2-D geometry {time.time() - t0:.1f} s", flush=True)

# ---- 3-D part on the device; torch shape (nz, ny, nx) is Julia's (nx, ny, nz) column-major ----
def dev2(a):  # (nx, ny) host -> (ny, nx) device view of the same column-major data
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)).to(dev)

d_depth, d_area = dev2(depth), dev2(area)
d_dz = torch.from_numpy(dz).to(dev)[:, None, None]
d_ztop = torch.from_numpy(zbot - dz).to(dev)[:, None, None]
thk = torch.minimum((d_depth[None] - d_ztop).clamp_min_(0.0), d_dz)
thk = torch.where(thk < 0.2 * d_dz, torch.zeros((), dtype=torch.float64, device=dev), thk)
wet = thk > 0
nanv = torch.full((), float("nan"), dtype=torch.float64, device=dev)
v3d = torch.where(wet, thk * d_area[None], nanv)            # volcello == 0 -> NaN (gridcellgeometry.jl:270-276)
thkc = torch.where(wet, thk, nanv)                           # thkcello = v3D / area2D
gen = torch.Generator(device=dev)
gen.manual_seed(20260501)
sig = 1.0e9 / float(dz.max())
fillv = torch.full((), FILL, dtype=torch.float64, device=dev)
umo = torch.where(wet, torch.randn(thk.shape, generator=gen, dtype=torch.float64, device=dev) * thk * sig, fillv)
vmo = torch.where(wet, torch.randn(thk.shape, generator=gen, dtype=torch.float64, device=dev) * thk * sig, fillv)
rho = torch.where(wet, 1025.0 + 0.004 * (torch.cumsum(thk, 0) - 0.5 * thk)
                  + 0.1 * torch.randn(thk.shape, generator=gen, dtype=torch.float64, device=dev), nanv)
ml = dev2(np.where(vol1[:, :, 0] > 0, mlotst, np.nan))
del thk, wet
torch.cuda.synchronize()
print(f"[{wl}] device 3-D fields ready, {torch.cuda.memory_allocated() / 1e9:.1f} GB allocated, {time.time() - t0:.1f} s", flush=True)

asm = DeviceAssembler(0)
asm.set_grid_tensors(shape=(nx, ny, nz), topology=int(gm.gridtopology.kind), v3d=v3d.reshape(-1), thkcello=thkc.reshape(-1),
                     edge_length=[dev2(gm.edge_length_2D[d]).reshape(-1) for d in HDIRS],
                     dist_nbr=[dev2(gm.distance_to_neighbour_2D[d]).reshape(-1) for d in HDIRS],
                     area2d=d_area.reshape(-1), zt=torch.from_numpy(zt).to(dev), mlotst=ml.reshape(-1), rho=rho.reshape(-1))
umo, vmo = umo.reshape(-1), vmo.reshape(-1)
N = asm.N
print(f"[{wl}] G={G} N={N} ({N / G:.3f} wet)", flush=True)

def one_step():
    asm.step(umo, vmo, FILL, onepass=False)  # plan -> exactly sized outputs -> fill

one_step()
asm.ctx.synchronize()
print(f"[{wl}] first step done, nnz={asm.nnz}, {torch.cuda.memory_allocated() / 1e9:.1f} GB allocated", flush=True)
t1 = time.perf_counter()
for _ in range(steps):
    one_step()
asm.ctx.synchronize()
ms = 1e3 * (time.perf_counter() - t1) / steps
asm.ctx.timing_enable(True)
for _ in range(steps):
    one_step()
kt = {k: v[0] / v[1] for k, v in asm.ctx.timing_collect().items()}
asm.ctx.timing_enable(False)
bytes_tm, bytes_ff = asm.algorithmic_bytes(), asm.facefluxes_bytes()
res = dict(workload=wl, G=G, N=N, nnz=dict(zip(MATS, asm.nnz)), protocol="plan+fill (host-synchronised twice per step)",
           ms_per_step=ms, wet_cells_per_s=N / (ms * 1e-3), kernels_ms=kt,
           tm_fill_GBs=bytes_tm / (kt["tm_kernel<fill>"] * 1e-3) / 1e9,
           facefluxes_GBs=bytes_ff / (kt["facefluxes_kernel"] * 1e-3) / 1e9, algorithmic_bytes=bytes_tm,
           device_GB_allocated=torch.cuda.memory_allocated() / 1e9)
print(json.dumps(res), flush=True)

# ---- properties, level by level / column chunk by column chunk ----
checks = {}
phi = [p.view(nz, ny, nx) for p in asm.phi]
e, w_, n_, s_, top, bot = phi
ok = dict(phi_finite=True, bottom_is_top_below=True, west_is_east_shifted=True, south_is_north_shifted=True, continuity_exact=True)
for k in range(nz):
    ok["phi_finite"] &= bool(all(torch.isfinite(p[k]).all() for p in phi))
    ok["bottom_is_top_below"] &= bool(torch.equal(bot[k], top[k + 1]) if k + 1 < nz else (bot[k] == 0).all())
    ok["west_is_east_shifted"] &= bool(torch.equal(w_[k], torch.roll(e[k], 1, dims=1)))
    ok["south_is_north_shifted"] &= bool(torch.equal(s_[k, 1:], n_[k, :-1]) and (s_[k, 0] == 0).all())
    ok["continuity_exact"] &= bool((((((bot[k] + w_[k]) + s_[k]) - e[k]) - n_[k]) - top[k] == 0).all())
checks.update(ok)
vw = asm.v3d[asm.lwet[:N] - 1]
x = torch.randn(N, dtype=torch.float64, device=dev)
Myr = 365.25 * 86400 * 1e6
NCH = 64
bounds = [N * c // NCH for c in range(NCH + 1)]
acc_Tx = torch.zeros(N, dtype=torch.float64, device=dev)
Tx = None
for kk, m in enumerate(MATS):
    cp, rv, nzv = asm.out[m]
    nn = asm.nnz[kk]
    wellformed = bool(cp[0] == 1 and cp[N] == nn + 1 and (cp[1:] >= cp[:-1]).all())
    rowsum = torch.zeros(N, dtype=torch.float64, device=dev)
    MTv = torch.zeros(N, dtype=torch.float64, device=dev)
    Mx = torch.zeros(N, dtype=torch.float64, device=dev)
    diag_ok, offd_ok, nozero, ndiag = True, True, True, 0
    for c in range(NCH):
        c0, c1 = bounds[c], bounds[c + 1]
        a, b = int(cp[c0]) - 1, int(cp[c1]) - 1
        if b <= a:
            continue
        r, z = rv[a:b], nzv[a:b]
        cnt = cp[c0 + 1:c1 + 1] - cp[c0:c1]
        col = torch.repeat_interleave(torch.arange(c0, c1, device=dev), cnt)
        first = torch.zeros(b - a, dtype=torch.bool, device=dev)
        first[(cp[c0:c1] - 1 - a)[cnt > 0]] = True
        wellformed &= bool(((r[1:] > r[:-1]) | first[1:]).all()) and bool((r >= 1).all() and (r <= N).all())
        rowsum.index_add_(0, r - 1, z)
        MTv.index_add_(0, col, z * vw[r - 1])
        Mx.index_add_(0, r - 1, z * x[col])
        if m == "T":
            isd = (r - 1) == col
            ndiag += int(isd.sum())
            diag_ok &= bool((z[isd] > 0).all())
            offd_ok &= bool((z[~isd] < 0).all())
            nozero &= bool((z != 0).all())
        del col, first, r, z
    checks[f"{m}_csc_wellformed"] = wellformed
    if m == "T":
        Tx = Mx
        checks["T_diag_positive"] = diag_ok and ndiag == N
        checks["T_offdiag_negative"] = offd_ok
        checks["T_no_stored_zero"] = nozero
    else:
        acc_Tx += Mx
    if m not in ("T", "Tadv"):
        checks[f"{m}_divergence_Myr"] = float(torch.ones(N, dtype=torch.float64, device=dev).norm() / rowsum.norm().clamp_min(1e-300) / Myr)
        checks[f"{m}_volume_Myr"] = float(vw.norm() / MTv.norm().clamp_min(1e-300) / Myr)
checks["T_is_sum_of_operators_relerr"] = float((Tx - acc_Tx).norm() / Tx.norm())
bad = [k for k, val in checks.items() if (val is False) or (k.endswith("_Myr") and val < 1e6) or (k.endswith("relerr") and val > 1e-12)]
print(json.dumps(dict(checks=checks, failed=bad)))
sys.exit(1 if bad else 0)
