#!/usr/bin/env python3
"""Probe: does running facefluxes of time slice s+1 on a second stream, concurrently with count/fill of slice s, raise the
throughput?  Two phi/mask buffer sets, events for the two dependencies.  python tools/overlap_probe.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
asm.step(umo, vmo, 1e20)
# baseline: one stream
for _ in range(10):
    asm.step_async(umo, vmo, 1e20)
asm.finish(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    asm.step_async(umo, vmo, 1e20)
asm.finish(); torch.cuda.synchronize()
base = (time.perf_counter() - t0) / steps
# two streams, two buffer sets
s_ff, s_tm = torch.cuda.Stream(), torch.cuda.Stream()
bufs = []
for b in range(2):
    bufs.append(([torch.empty(asm.G, dtype=torch.float64, device="cuda") for _ in range(6)], torch.empty(asm.G, dtype=torch.int16, device="cuda")))
ff_done = [torch.cuda.Event() for _ in range(2)]
tm_done = [torch.cuda.Event() for _ in range(2)]
def run(n):
    for s in range(n):
        b = s & 1
        asm.phi, asm.push_mask = bufs[b]
        asm.ctx.set_stream(s_ff.cuda_stream)
        if s >= 2:
            s_ff.wait_event(tm_done[b])          # the transportmatrix that read this buffer set two slices ago
        phi = asm.facefluxes_async(umo, vmo, 1e20)
        ff_done[b].record(s_ff)
        asm.ctx.set_stream(s_tm.cuda_stream)
        s_tm.wait_event(ff_done[b])
        asm.transportmatrix_onepass(phi, sync=False)
        tm_done[b].record(s_tm)
    torch.cuda.synchronize()
run(10)
t0 = time.perf_counter()
run(steps)
over = (time.perf_counter() - t0) / steps
asm.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
print(f"one stream {1e3 * base:.4f} ms/step   two streams (facefluxes overlapped) {1e3 * over:.4f} ms/step   ratio {over / base:.3f}")
