#!/usr/bin/env python3
# NOTE (round 6): needs the dense-march formulation, which left the product: apply tools/experiments/r06_removed_formulations.patch first.
"""Variants of the library (built beforehand: tools/dense_ab.py --build name="flags" ...) timed one after the other on one
device-generated grid:   python tools/dense_ab.py --run name1,name2 [--workload quarterdeg] [--dense 1] [--kparts 1]"""
import argparse
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(pairs):
    spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for arg in pairs:
        name, flags = arg.split("=", 1)
        print(b.build(force=True, extra=flags.split(), name=name))


def run(names, workload, dense, kparts, steps, rows=0):
    import torch

    from otmb_amd import capi, synthetic_device

    dev = torch.device("cuda", 0)
    dg = synthetic_device.make_device_grid(workload, dev, seed=20260501, rho="array")
    ref = None
    for name in names:
        path = os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "lib", "libotmb_hip.so" if name == "default" else f"libotmb_hip_{name}.so")
        capi.use_library(path, lenient=True)
        asm = synthetic_device.assembler_for(dg, 0)
        asm.ctx.set_formulation(dense, kparts)
        asm.ctx.set_tile_order(rows)
        for _ in range(4):
            asm.step_async(dg.umo, dg.vmo, dg.fill)
        asm.finish()
        asm.ctx.timing_enable(True)
        for _ in range(steps):
            asm.step_async(dg.umo, dg.vmo, dg.fill)
        asm.finish()
        kt = asm.ctx.timing_collect()
        chk = tuple((float(asm.out[m][2][: asm.nnz[q]].sum().item()), int(asm.nnz[q])) for q, m in enumerate(("T", "Tadv", "TκH", "TκVML", "TκVdeep")))
        if ref is None:
            ref = chk
        k = {n: round(v[0] / v[1], 4) for n, v in kt.items()}
        print(json.dumps({"variant": name, "rows": rows, "same_results": chk == ref, "kernels_ms": k}), flush=True)
        del asm
        torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", nargs="*")
    ap.add_argument("--run")
    ap.add_argument("--workload", default="quarterdeg")
    ap.add_argument("--dense", type=int, default=1)
    ap.add_argument("--kparts", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--rows", type=int, default=0)
    a = ap.parse_args()
    if a.build:
        build(a.build)
    if a.run:
        run(a.run.split(","), a.workload, a.dense, a.kparts, a.steps, a.rows)
