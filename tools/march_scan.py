#!/usr/bin/env python3
"""Fill-pass time against the tile order (otmb_ctx_set_tile_order), one process, one library, interleaved rounds:
    python tools/march_scan.py [--workload quarterdeg] [--rows 0,4,8,16,32] [--rounds 3] [--steps 10]
Prints the average HIP-event duration of every kernel per setting (the pool's boxes differ: only in-process ratios count)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="quarterdeg")
    ap.add_argument("--rows", default="0,4,8,16,32")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
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
    rows = [int(r) for r in args.rows.split(",")]
    for _ in range(5):
        asm.step_async(umo, vmo, fill)
    asm.finish()
    ref = None
    acc = {r: {} for r in rows}
    for rnd in range(args.rounds):
        for r in rows:
            asm.ctx.set_tile_order(r)
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
                a = acc[r].setdefault(k, [0.0, 0])
                a[0] += ms
                a[1] += n
            # results do not depend on the order: compare T's values with the first setting's
            t = asm.out["T"][2][: asm.nnz[0]]
            chk = (float(t.sum().item()), int(asm.nnz[0]))
            if ref is None:
                ref = chk
            assert chk == ref, (chk, ref)
    alg = asm.algorithmic_bytes()
    for r in rows:
        k = {n: v[0] / v[1] for n, v in acc[r].items()}
        f = k.get("tm_kernel<fill>", float("nan"))
        print(json.dumps({"workload": args.workload, "rows_per_band": r, "fill_ms": round(f, 4), "fill_TBs_algorithmic": round(alg / f / 1e9, 3),
                          "frac_of_8TBs": round(alg / f / 1e9 / 8, 3), "kernels_ms": {n: round(v, 4) for n, v in k.items()}}), flush=True)


if __name__ == "__main__":
    main()
