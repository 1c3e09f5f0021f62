#!/usr/bin/env python3
"""Full-size run of the hot path with size-independent property checks (no oracle needed):
   python tools/large_run.py quarterdeg|access1deg|tenthdeg [steps]
Checks on the device-resident result: CSC well-formedness (colptr, strictly ascending rows), T = sum of the
four operators (as operators on a random vector), T·1 ~ 0 for the diffusive operators and Tᵀv ~ 0 for all
(test/online.jl:110-115), diag(T) > 0 and off-diag < 0 for upwind (test/online.jl:119-123), facefluxes
identities (bottom[k] == top[k+1], west == shifted east, mass balance)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
from otmb_amd.capi import MATS

wl = sys.argv[1] if len(sys.argv) > 1 else "quarterdeg"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nx, ny, nz, lf = synthetic.PRESETS[wl]
t0 = time.time()
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
print(f"[{wl}] host grid generation {time.time() - t0:.1f} s", flush=True)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
dev = asm.device
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
N, G, P = asm.N, asm.G, nx * ny
print(f"[{wl}] G={G} N={N} ({N / G:.3f} wet)", flush=True)

asm.step(umo, vmo, 1e20)
asm.ctx.synchronize()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    asm.step(umo, vmo, 1e20)
asm.ctx.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
asm.ctx.timing_enable(True)
for _ in range(steps):
    asm.step(umo, vmo, 1e20)
kt = {k: v[0] / v[1] for k, v in asm.ctx.timing_collect().items()}
asm.ctx.timing_enable(False)
bytes_tm, bytes_ff = asm.algorithmic_bytes(), asm.facefluxes_bytes()
res = dict(workload=wl, G=G, N=N, nnz=dict(zip(MATS, asm.nnz)), ms_per_step=ms, wet_cells_per_s=N / (ms * 1e-3),
           kernels_ms=kt, tm_fill_GBs=bytes_tm / (kt["tm_kernel<fill>"] * 1e-3) / 1e9,
           facefluxes_GBs=bytes_ff / (kt["facefluxes_kernel"] * 1e-3) / 1e9, algorithmic_bytes=bytes_tm)

# ---- properties ----
checks = {}
phi = asm.phi
e, w_, n_, s_, top, bot = [p.view(nz, ny, nx) for p in phi]  # torch view: (k, j, i)
checks["phi_finite"] = bool(all(torch.isfinite(p).all() for p in phi))
checks["bottom_is_top_below"] = bool(torch.equal(bot[:-1], top[1:]) and (bot[-1] == 0).all())
checks["west_is_east_shifted"] = bool(torch.equal(w_, torch.roll(e, 1, dims=2)))
checks["south_is_north_shifted"] = bool(torch.equal(s_[:, 1:, :], n_[:, :-1, :]) and (s_[:, 0, :] == 0).all())
resid = (((bot + w_) + s_) - e) - n_ - top
checks["continuity_exact"] = bool((resid == 0).all())
v = asm.v3d[asm.lwet[:N] - 1]
ones = torch.ones(N, dtype=torch.float64, device=dev)
x = torch.randn(N, dtype=torch.float64, device=dev)
Myr = 365.25 * 86400 * 1e6
acc_Tx = torch.zeros(N, dtype=torch.float64, device=dev)
Tx = None
for k, m in enumerate(MATS):
    cp, rv, nzv = asm.out[m]
    nn = asm.nnz[k]
    rv, nzv = rv[:nn], nzv[:nn]
    ok = bool(cp[0] == 1 and cp[N] == nn + 1 and (cp[1:] >= cp[:-1]).all())
    # strictly ascending rows inside columns: a descent is allowed only where a new column starts
    starts = torch.zeros(nn + 1, dtype=torch.bool, device=dev)
    starts[(cp[:-1] - 1)[cp[:-1] <= nn]] = True
    asc = (rv[1:] > rv[:-1]) | starts[1:nn]
    ok = ok and bool(asc.all()) and bool((rv >= 1).all() and (rv <= N).all())
    checks[f"{m}_csc_wellformed"] = ok
    col = torch.repeat_interleave(torch.arange(N, device=dev), cp[1:] - cp[:-1])
    rowsum = torch.zeros(N, dtype=torch.float64, device=dev).index_add_(0, rv - 1, nzv)  # M·1
    MTv = torch.zeros(N, dtype=torch.float64, device=dev).index_add_(0, col, nzv * v[rv - 1])  # Mᵀv
    Mx = torch.zeros(N, dtype=torch.float64, device=dev).index_add_(0, rv - 1, nzv * x[col])
    if m == "T":
        Tx = Mx
        diag = nzv[(rv - 1) == col]
        checks["T_diag_positive"] = bool(diag.numel() == N and (diag > 0).all())
        checks["T_offdiag_negative"] = bool((nzv[(rv - 1) != col] < 0).all())
        checks["T_no_stored_zero"] = bool((nzv != 0).all())
    else:
        acc_Tx += Mx
    if m not in ("T", "Tadv"):
        checks[f"{m}_divergence_Myr"] = float(ones.norm() / rowsum.norm().clamp_min(1e-300) / Myr)
    if m not in ("T", "Tadv"):  # with a 3-D rho the advective operator conserves mass, not volume
        checks[f"{m}_volume_Myr"] = float(v.norm() / MTv.norm().clamp_min(1e-300) / Myr)
    del col, rowsum, MTv, Mx
checks["T_is_sum_of_operators_relerr"] = float((Tx - acc_Tx).norm() / Tx.norm())
res["checks"] = checks
bad = [k for k, val in checks.items() if (val is False) or (k.endswith("_Myr") and val < 1e6) or (k.endswith("relerr") and val > 1e-12)]
res["failed"] = bad
print(json.dumps(res))
sys.exit(1 if bad else 0)
