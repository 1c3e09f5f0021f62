#!/usr/bin/env python3
"""Can a per-ARRAY choice beat the best whole output set?  0.25 degree grid, NSETS output sets; start from the fastest set and, array by array (rowval / nzval of the
five matrices), try the same array of every other set in its place; keep a swap that makes the fill pass faster.  Prints the trajectory.
    python tools/placement_greedy.py [NSETS]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from otmb_amd import synthetic_device
from otmb_amd.capi import MATS

nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
dg = synthetic_device.make_device_grid("quarterdeg", dev, seed=20260501, rho="array")
asm = synthetic_device.assembler_for(dg, 0)
phi = asm.facefluxes(dg.umo, dg.vmo, dg.fill)
sets = [asm.new_output_set() for _ in range(nsets)]


def fill_ms(out, reps=3):
    for _ in range(2):
        asm.transportmatrix_onepass(phi, sync=False, out=out)
    asm.result()
    asm.ctx.synchronize()
    asm.ctx.timing_enable(True)
    for _ in range(reps):
        asm.transportmatrix_onepass(phi, sync=False, out=out)
    asm.result()
    asm.ctx.synchronize()
    t = asm.ctx.timing_collect()
    asm.ctx.timing_enable(False)
    return t["tm_kernel<fill>"][0] / t["tm_kernel<fill>"][1]


base = [round(fill_ms(s), 4) for s in sets]
best = int(np.argmin(base))
cur = {m: list(sets[best][m]) for m in MATS}
cur_ms = fill_ms({m: tuple(v) for m, v in cur.items()})
traj = [{"start": best, "ms": round(cur_ms, 4)}]
for m in MATS:
    for q in (1, 2):  # rowval, nzval
        for k in range(nsets):
            if k == best:
                continue
            trial = {mm: list(v) for mm, v in cur.items()}
            trial[m][q] = sets[k][m][q]
            ms = fill_ms({mm: tuple(v) for mm, v in trial.items()})
            if ms < cur_ms * 0.997:
                cur, cur_ms = trial, ms
                traj.append({"array": f"{m}[{q}]", "from_set": k, "ms": round(ms, 4)})
print(json.dumps({"whole_sets_ms": base, "trajectory": traj, "final_ms": round(fill_ms({m: tuple(v) for m, v in cur.items()}), 4)}))
