#!/usr/bin/env python3
"""Where does a wave of the fill pass spend its life?  Diagnostic build (-DOTMB_DBG_STAMPS: s_memtime stamps in SGPRs,
written by lane 0 of each wave to a buffer of their own; fences around every stamp forbid overlaps the real kernel has,
so read the SHARES, not the length).   gpurun -- python tools/stamps.py [workload] [extra -D flags]
Stamps: 0 entry | 1 Lwet back | 2 stencil loads back | 3 arithmetic done | 4 tile offsets known | 5 stores issued | 6 stores acked"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
extra = sys.argv[2:]
path = b.build(force=True, extra=["-DOTMB_DBG_STAMPS", *extra], name="stamps")
import torch
import otmb_amd
from otmb_amd import capi, synthetic
from otmb_amd.device import DeviceAssembler

capi.use_library(path, lenient=True)
nx, ny, nz, lf = synthetic.PRESETS[wl]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
for _ in range(5):
    asm.step(umo, vmo, 1e20)
asm.ctx.timing_enable(True)
for _ in range(5):
    asm.step(umo, vmo, 1e20)
print({k: round(v[0] / v[1], 4) for k, v in asm.ctx.timing_collect().items()})
ntiles = (asm.N + 255) // 256
nw = ntiles * 4
buf = np.zeros(nw * 8, dtype=np.uint64)
fn = capi.lib().otmb_debug_stamps
fn.restype = C.c_int32
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = fn(asm.ctx.handle, buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.reshape(nw, 8).astype(np.int64)
ok = (t[:, :7] > 0).all(axis=1)
t = t[ok]
print(f"{wl}: {nw} waves, {ok.sum()} with all stamps")
names = ["entry->Lwet back", "->stencil loads back", "->arithmetic done", "->offsets known (scan+barrier)", "->stores issued (5 x stage+store)",
         "->stores acked"]
if "-DOTMB_DBG_STAMPS_ORDER" in extra:  # slot 6 = "tile id known" instead of "stores acked"
    e = t[:, 6] - t[:, 0]
    print(f"  entry->tile id known (kernel arguments, tile order) mean {e.mean():9.0f}  median {np.median(e):9.0f}  p90 {np.percentile(e, 90):9.0f}")
    t[:, 6] = t[:, 5]
d = np.diff(t[:, :7], axis=1)
life = t[:, 6] - t[:, 0]
for q, n in enumerate(names):
    print(f"  {n:40s} mean {d[:, q].mean():9.0f}  median {np.median(d[:, q]):9.0f}  p90 {np.percentile(d[:, q], 90):9.0f}  share {d[:, q].sum() / life.sum():6.1%}")
print(f"  wave life: mean {life.mean():.0f} median {np.median(life):.0f} cycles (s_memtime ticks)")
span = t[:, 6].max() - t[:, 0].min()
print(f"  kernel span {span} ticks; sum of wave lives / span = {life.sum() / span:.1f} waves in flight on average ({life.sum() / span / 256:.2f} per CU)")
# start-time profile: how many dispatch rounds
st = np.sort(t[:, 0] - t[:, 0].min())
print("  wave start times (ticks) deciles:", [int(x) for x in np.percentile(st, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])])
hw = t[:, 7]
cu_key = (hw >> 8) & 0xfffff  # everything above the wave-slot/SIMD bits: CU, SH, SE, ...
print("  distinct HW_ID>>8 values:", len(np.unique(cu_key)))
