#!/usr/bin/env python3
"""Same number of cells per level, different row length: does the per-cell cost depend on nx?
   gpurun -- python tools/shape_scan.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import otmb_amd
from otmb_amd import synthetic_device

dev = torch.device("cuda", 0)
import os as _os
SHAPES = eval(_os.environ.get("SHAPES", "((90, 1200, 50), (180, 600, 50), (360, 300, 50), (720, 150, 50), (1440, 76, 50), (2880, 38, 50))"))
for nx, ny, nz in SHAPES:
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
    print(f"{nx:5d} x {ny:5d} x {nz} N={N:9d} ({N / asm.G:.3f} wet) fill {fill:.4f} ms  {1e6 * fill / N:.4f} ns/cell  {asm.algorithmic_bytes() / fill / 1e6:.0f} GB/s | count "
          f"{kt['tm_count_kernel']:.4f} | facefluxes {kt['facefluxes_kernel']:.4f} ({asm.facefluxes_bytes() / kt['facefluxes_kernel'] / 1e6:.0f} GB/s)", flush=True)
    del asm, dg
    torch.cuda.empty_cache()
