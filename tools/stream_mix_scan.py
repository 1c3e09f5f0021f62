#!/usr/bin/env python3
"""What an ideal streaming kernel reaches over the fill pass's OWN arrays as a function of the slice length (otmb_ctx_stream_mix with
N / 256 ... N / 8192 tiles; N / 256 is the fill pass's own granularity), beside the fill pass itself.  profiles/r05/README.md.
    python tools/stream_mix_scan.py [workload]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from otmb_amd import synthetic_device
from otmb_amd.capi import MATS

wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
dev = torch.device("cuda", 0)
dg = synthetic_device.make_device_grid(wl, dev, seed=20260501, rho="array")
asm = synthetic_device.assembler_for(dg, 0)
for _ in range(5):
    asm.step_async(dg.umo, dg.vmo, dg.fill)
asm.finish()
asm.ctx.timing_enable(True)
for _ in range(10):
    asm.step_async(dg.umo, dg.vmo, dg.fill)
asm.finish()
kt = asm.ctx.timing_collect()
asm.ctx.timing_enable(False)
fill_ms = kt["tm_kernel<fill>"][0] / kt["tm_kernel<fill>"][1]
alg = asm.algorithmic_bytes()
b8 = lambda t, n=None: (t.data_ptr(), 8 * (t.numel() if n is None else n))
ins = [b8(p) for p in asm.phi] + [b8(asm.v3d), b8(asm.thk), b8(asm.lwet3d)] + ([b8(asm.rho)] if asm.rho is not None else [])
ins += [b8(t) for t in (*asm.edge, *asm.dist, asm.area, asm.mlotst)]
outs = []
for k, m in enumerate(MATS):
    cp, rv, nz = asm.out[m]
    outs += [b8(cp, asm.N + 1), b8(rv, asm.nnz[k]), b8(nz, asm.nnz[k])]
rec = {"workload": wl, "fill_ms": fill_ms, "fill_gbs": alg / (fill_ms * 1e-3) / 1e9, "mix_gbs_by_columns_per_slice": {}}
for cols in (64, 128, 256, 512, 1024, 2048, 4096, 16384):
    rec["mix_gbs_by_columns_per_slice"][cols] = round(asm.ctx.stream_mix(ins, outs, max(8, asm.N // cols)), 1)
rec["reads_only_gbs_256"] = round(asm.ctx.stream_mix(ins, [], max(8, asm.N // 256)), 1)
rec["writes_only_gbs_256"] = round(asm.ctx.stream_mix([], outs, max(8, asm.N // 256)), 1)
print(json.dumps(rec), flush=True)
