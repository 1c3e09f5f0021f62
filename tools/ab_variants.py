#!/usr/bin/env python3
"""In-process A/B of library variants (box-to-box variance is ~15 %, so variants must share a process):
   python tools/ab_variants.py base="" nt="-DOTMB_NT_STORES" ...      [WORKLOAD=access1deg ROUNDS=4]
Each name=flags pair is built as lib/libotmb_hip_<name>.so, then rounds of 10 steps are interleaved."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
variants = {}
for arg in sys.argv[1:]:
    name, flags = arg.split("=", 1)
    if flags.startswith("@"):  # a prebuilt library (e.g. an older commit built with tools/build_ref_variant.sh)
        variants[name] = os.path.join(ROOT, flags[1:])
    else:
        variants[name] = b.build(force=True, extra=flags.split(), name=name)
import torch
import otmb_amd
from otmb_amd import capi, synthetic
from otmb_amd.device import DeviceAssembler
wl = os.environ.get("WORKLOAD", "access1deg")
nx, ny, nz, lf = synthetic.PRESETS[wl]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
# The placement of the arrays in HBM moves the fill pass by +-5 % (tools/placement_study.py) and later allocations tend
# to be slower, so every variant gets REPS assemblers, created in interleaved order; the figure is the mean over its
# assemblers of the per-assembler median.
reps = int(os.environ.get("REPS", "3"))
asms = []
for rep in range(reps):
    order = list(variants.items())
    if rep % 2:
        order.reverse()
    for name, path in order:
        capi.use_library(path, lenient=True)
        a = DeviceAssembler(0)
        a.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
        try:
            a.step(umo, vmo, 1e20)
        except capi.OtmbError as e:  # debug variants that fake inputs may trip the library's checks; timings still count
            print(name, "->", e)
        a.ctx.timing_enable(True)
        asms.append((name, a))
res = {}
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for i, (name, a) in enumerate(asms):
        for _ in range(10):
            try:
                a.step(umo, vmo, 1e20)
            except capi.OtmbError:
                pass
        for k, v in a.ctx.timing_collect().items():
            res.setdefault(name, {}).setdefault(k, {}).setdefault(i, []).append(v[0] / v[1])
for name in variants:
    out = {}
    for k, d in res[name].items():
        if "finish" in k or "mask" in k:
            continue
        per = [float(np.median(v)) for v in d.values()]
        out[k.replace("_kernel", "").replace("tm_kernel", "tm")] = f"{np.mean(per):.4f} [{min(per):.4f}-{max(per):.4f}]"
    print(f"{name:10s}", out)
