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
asms = {}
for name, path in variants.items():
    capi.use_library(path, lenient=True)
    a = DeviceAssembler(0)
    a.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    try:
        a.step(umo, vmo, 1e20)
    except capi.OtmbError as e:  # debug variants that fake inputs may trip the library's checks; timings still count
        print(name, "->", e)
    a.ctx.timing_enable(True)
    asms[name] = a
res = {n: {} for n in asms}
for rnd in range(int(os.environ.get("ROUNDS", "4"))):
    for name, a in asms.items():
        for _ in range(10):
            try:
                a.step(umo, vmo, 1e20)
            except capi.OtmbError:
                pass
        for k, v in a.ctx.timing_collect().items():
            res[name].setdefault(k, []).append(v[0] / v[1])
for name, r in res.items():
    print(f"{name:12s}", {k: f"{np.median(v):.4f} (min {min(v):.4f})" for k, v in r.items() if "finish" not in k})
