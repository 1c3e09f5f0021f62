#!/usr/bin/env python3
"""Extension timing: T alone (otmb_tm_args.only_t) against the reference behaviour (all five matrices), 1 degree grid."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
nx, ny, nz, lf = synthetic.PRESETS[wl]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
for only in (False, True, False, True):
    asm = DeviceAssembler(0)
    asm.only_T = only
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    for _ in range(10):
        asm.step_async(umo, vmo, 1e20)
    asm.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        asm.step_async(umo, vmo, 1e20)
    asm.finish(); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 50
    asm.ctx.timing_enable(True)
    for _ in range(10):
        asm.step_async(umo, vmo, 1e20)
    asm.finish()
    kt = {k: round(v[0] / v[1], 4) for k, v in asm.ctx.timing_collect().items()}
    print(f"{wl} only_T={only}: {ms:.4f} ms/step = {asm.N / ms * 1e3:.3e} wet-cells/s  nnz={asm.nnz}  {kt}", flush=True)
