#!/usr/bin/env python3
# NOTE (round 6): needs the dense-march formulation, which left the product: apply tools/experiments/r06_removed_formulations.patch first.
"""Gather kernels against the dense-tile march (otmb_ctx_set_formulation), one process, interleaved rounds:
    python tools/dense_scan.py [--workload quarterdeg] [--settings g,d1,d2,d3,d5] [--rounds 2] [--steps 8]
g = gather (wet-rank tile order), gN = gather in march order with bands of N rows, dK = dense march with K depth parts."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="quarterdeg")
    ap.add_argument("--settings", default="g,d1,d2,d3,d5")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--rho", default="array")
    args = ap.parse_args()
    import numpy as np
    import torch

    import otmb_amd
    from otmb_amd import synthetic, synthetic_device
    from otmb_amd.device import DeviceAssembler

    dev = torch.device("cuda", 0)
    if args.workload in ("quarterdeg", "tenthdeg"):
        dg = synthetic_device.make_device_grid(args.workload, dev, seed=20260501, rho=args.rho)
        asm = synthetic_device.assembler_for(dg, 0)
        umo, vmo, fill = dg.umo, dg.vmo, dg.fill
    else:
        nx, ny, nz, lf = synthetic.PRESETS[args.workload]
        g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho=args.rho)
        gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                      lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
        asm = DeviceAssembler(0)
        asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=True)
        umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
        vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
        fill = g.umo.properties["_FillValue"]
    settings = args.settings.split(",")

    def apply(s):
        if s[0] == "g":
            asm.ctx.set_formulation(0)
            asm.ctx.set_tile_order(int(s[1:]) if len(s) > 1 else 0)
        else:
            asm.ctx.set_formulation(1, int(s[1:]) if len(s) > 1 else 1)

    ref = None
    acc = {s: {} for s in settings}
    for rnd in range(args.rounds):
        for s in settings:
            apply(s)
            for _ in range(3):
                asm.step_async(umo, vmo, fill)
            asm.finish()
            asm.ctx.synchronize()
            asm.ctx.timing_enable(True)
            for _ in range(args.steps):
                asm.step_async(umo, vmo, fill)
            asm.finish()
            kt = asm.ctx.timing_collect()
            asm.ctx.timing_enable(False)
            for k, (ms, n) in kt.items():
                a = acc[s].setdefault(k, [0.0, 0])
                a[0] += ms
                a[1] += n
            chk = tuple((float(asm.out[m][2][: asm.nnz[q]].sum().item()), int(asm.out[m][1][: asm.nnz[q]].sum().item()), int(asm.out[m][0].sum().item()),
                         int(asm.nnz[q])) for q, m in enumerate(("T", "Tadv", "TκH", "TκVML", "TκVdeep")))
            if ref is None:
                ref = chk
            assert chk == ref, (s, chk, ref)
    alg = asm.algorithmic_bytes()
    for s in settings:
        k = {n: v[0] / v[1] for n, v in acc[s].items()}
        f = k.get("dm_fill_kernel", k.get("tm_kernel<fill>", float("nan")))
        tm = sum(v for n, v in k.items() if n != "facefluxes_kernel")
        print(json.dumps({"workload": args.workload, "setting": s, "fill_ms": round(f, 4), "transportmatrix_ms": round(tm, 4),
                          "fill_TBs_algorithmic": round(alg / f / 1e9, 3), "frac_of_8TBs": round(alg / f / 1e9 / 8, 3),
                          "kernels_ms": {n: round(v, 4) for n, v in k.items()}}), flush=True)


if __name__ == "__main__":
    main()
