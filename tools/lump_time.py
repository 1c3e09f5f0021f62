#!/usr/bin/env python3
"""Time lump_and_spray at full size: device (resident T) vs the oracle on the host.  python tools/lump_time.py [workload]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import otmb_amd
from otmb_amd import synthetic
from otmb_amd.device import DeviceAssembler
from oracle import oracle as orc
wl = sys.argv[1] if len(sys.argv) > 1 else "access1deg"
nx, ny, nz, lf = synthetic.PRESETS[wl]
g = synthetic.make_grid(nx, ny, nz, land_fraction=lf, rho="array")
gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                              lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
asm = DeviceAssembler(0)
asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
asm.step(umo, vmo, 1e20)
host = asm.result_to_host()
wet = (asm.wet3d.cpu().numpy() != 0).reshape((nx, ny, nz), order="F")
vol = np.asarray(gm.v3D).reshape(-1, order="F")[wet.reshape(-1, order="F")]
lat = np.asarray(g.lat)
region = np.repeat((lat > -35)[:, :, None], nz, axis=2)  # like test/online.jl:126-129: no lumping south of 35S
orc.build()
for name, mask, (di, dj, dk) in (("all, 2x2x1", None, (2, 2, 1)), ("region, 2x2x1", region, (2, 2, 1)), ("region, 10x10x1", region, (10, 10, 1)),
                                 ("all, 2x2x2", None, (2, 2, 2))):
    dm = None if mask is None else torch.from_numpy(np.asfortranarray(mask).ravel(order="F").astype(np.uint8)).cuda()
    asm.lump_and_spray(dm, di, dj, dk); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        L, S, vc = asm.lump_and_spray(dm, di, dj, dk)
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    rL, rS, rvc = orc.lump_and_spray(wet, vol, host["T"], mask, di, dj, dk)
    tc = time.perf_counter() - t0
    same = (np.array_equal(L[1].cpu().numpy(), rL[1]) and np.array_equal(L[2].cpu().numpy(), rL[2]) and np.array_equal(S[0].cpu().numpy(), rS[0])
            and np.array_equal(S[1].cpu().numpy(), rS[1]) and np.array_equal(vc.cpu().numpy(), rvc))
    print(f"{wl} {name:18s} N={asm.N} -> Nc={len(rvc)}  device {1e3 * tg:8.2f} ms   host oracle {1e3 * tc:8.1f} ms   bit-identical={same}", flush=True)
