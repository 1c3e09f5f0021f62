#!/usr/bin/env python3
"""Fixed overhead vs per-cell cost of the kernels: the 1 degree horizontal grid with 12..400 levels (device-generated).
   gpurun -- python tools/size_scan.py [nx ny]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import otmb_amd
from otmb_amd import synthetic_device

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (360, 300)
dev = torch.device("cuda", 0)
rows = []
for nz in (12, 25, 50, 100, 200, 400):
    dg = synthetic_device.make_device_grid((nx, ny, nz), dev)
    asm = synthetic_device.assembler_for(dg)
    for _ in range(3):
        asm.step_async(dg.umo, dg.vmo, dg.fill)
    asm.finish()
    asm.ctx.timing_enable(True)
    for _ in range(10):
        asm.step_async(dg.umo, dg.vmo, dg.fill)
    asm.finish()
    kt = {k: v[0] / v[1] for k, v in asm.ctx.timing_collect().items()}
    asm.ctx.timing_enable(False)
    N = asm.N
    fill = kt["tm_kernel<fill>"]
    rows.append((nz, N, fill, kt["tm_count_kernel"], kt["facefluxes_kernel"]))
    print(f"nz={nz:4d} N={N:9d} fill {fill:.4f} ms  {1e6 * fill / N:.4f} ns/cell  {asm.algorithmic_bytes() / fill / 1e6:.0f} GB/s | count {kt['tm_count_kernel']:.4f} "
          f"| facefluxes {kt['facefluxes_kernel']:.4f} ({asm.facefluxes_bytes() / kt['facefluxes_kernel'] / 1e6:.0f} GB/s)", flush=True)
    del asm, dg
    torch.cuda.empty_cache()
N = np.array([r[1] for r in rows], float)
for name, col in (("fill", 2), ("count", 3), ("facefluxes", 4)):
    t = np.array([r[col] for r in rows])
    a, b = np.polyfit(N, t, 1)
    print(f"{name}: t = {b * 1e3:.1f} us + {a * 1e6:.4f} ns/cell * N")
