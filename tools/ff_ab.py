#!/usr/bin/env python3
"""facefluxes with five wet bytes per level (otmb_facefluxes_slab_dev) against the folded flags byte (otmb_facefluxes_flags_dev),
one process, interleaved:  python tools/ff_ab.py [--workload access1deg]"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="access1deg"); a = ap.parse_args()
import numpy as np, torch
import otmb_amd
from otmb_amd import capi, synthetic, synthetic_device
from otmb_amd.device import DeviceAssembler
dev = torch.device("cuda", 0)
if a.workload in ("quarterdeg", "tenthdeg"):
    dg = synthetic_device.make_device_grid(a.workload, dev); asm = synthetic_device.assembler_for(dg, 0); umo, vmo, fill = dg.umo, dg.vmo, dg.fill
else:
    nx, ny, nz, lf = synthetic.PRESETS[a.workload]
    g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    asm = DeviceAssembler(0); asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev); vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev); fill = 1e20
asm.facefluxes(umo, vmo, fill)
ptrs = capi.ptr_array(6, [p.data_ptr() for p in asm.phi])
def run(flags, n=50):
    fn = asm.lib.otmb_facefluxes_flags_dev if flags else asm.lib.otmb_facefluxes_slab_dev
    w = asm.wetflags if flags else asm.wet3d
    for _ in range(n):
        asm.ctx.check(fn(asm.ctx.handle, umo.data_ptr(), vmo.data_ptr(), 0, w.data_ptr(), float(fill), asm.nx, asm.ny, asm.nz, asm.topology, C.byref(ptrs), None, asm.push_mask.data_ptr()))
res = {False: [], True: []}
ref = None
for rnd in range(6):
    for fl in (False, True):
        run(fl, 10); asm.ctx.synchronize(); asm.ctx.timing_enable(True); run(fl); kt = asm.ctx.timing_collect(); asm.ctx.timing_enable(False)
        res[fl].append(kt["facefluxes_kernel"][0] / kt["facefluxes_kernel"][1])
        chk = tuple(float(p.sum().item()) for p in asm.phi) + (int(asm.push_mask.to(torch.int64).sum().item()),)
        ref = ref or chk; assert chk == ref
print(json.dumps({"workload": a.workload, "five_wet_bytes_ms": round(float(np.median(res[False])), 5), "flags_byte_ms": round(float(np.median(res[True])), 5)}))
