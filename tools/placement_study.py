#!/usr/bin/env python3
"""How much does the placement of the arrays in HBM move the kernel times?  Several assemblers per setting (each with its
own allocations), interleaved rounds, HIP-event kernel times.  OTMB_STAGGER=<bytes> offsets the k-th array by k*bytes.
   python tools/placement_study.py 0 4352 135424     [WORKLOAD=access1deg REPS=4]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
wl = os.environ.get("WORKLOAD", "access1deg")
reps = int(os.environ.get("REPS", "4"))
nx, ny, nz, lf = synthetic.PRESETS[wl]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
hu = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F"))
hv = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F"))
asms = []
for rep in range(reps):
    for st in sys.argv[1:]:
        os.environ["OTMB_STAGGER"] = st
        a = DeviceAssembler(0)
        a.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
        umo, vmo = a._t(hu.numpy()), a._t(hv.numpy())
        a.step(umo, vmo, 1e20)
        a.ctx.timing_enable(True)
        asms.append((st, a, umo, vmo))
        junk = torch.empty(int(37e6) * (rep + 1), dtype=torch.uint8, device="cuda")  # shift the next allocations
res = {}
for rnd in range(4):
    for i, (st, a, umo, vmo) in enumerate(asms):
        for _ in range(10):
            a.step(umo, vmo, 1e20)
        for k, v in a.ctx.timing_collect().items():
            res.setdefault((st, k), {}).setdefault(i, []).append(v[0] / v[1])
for (st, k), d in sorted(res.items()):
    if "fill" in k or "facefluxes" in k:
        per = [float(np.median(v)) for v in d.values()]
        print(f"stagger {st:>8s} {k:20s} per-assembler medians: " + " ".join(f"{x:.4f}" for x in per) + f"   mean {np.mean(per):.4f}")
