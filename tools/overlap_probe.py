#!/usr/bin/env python3
"""Does the chip gain anything when the steps of TWO assemblers (two contexts = two HIP streams) run side by side?  An upper bound for
what overlapping one time slice's facefluxes / counting pass with the previous slice's fill pass could buy.
   python tools/overlap_probe.py [workload] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from otmb_amd import synthetic_device

wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
dg = synthetic_device.make_device_grid(wl, dev, seed=20260501, rho="array")
a = synthetic_device.assembler_for(dg, 0)
b = synthetic_device.assembler_for(dg, 0)


def run(asms, n):
    for x in asms:
        for _ in range(3):
            x.step_async(dg.umo, dg.vmo, dg.fill)
        x.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for x in asms:
            x.step_async(dg.umo, dg.vmo, dg.fill)
    for x in asms:
        x.finish()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (n * len(asms))


for rep in range(3):
    one = run([a], steps)
    two = run([a, b], steps // 2)
    print(f"{wl}: one stream {one:.4f} ms per step, two streams side by side {two:.4f} ms per step ({one / two:.3f} x)", flush=True)
