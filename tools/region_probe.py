#!/usr/bin/env python3
"""Does the fill pass's time at 1 degree depend on WHERE in the HBM the process's arrays lie?  One process; before each trial a dummy block of
a given size is allocated (and kept) so that the assembler's ~3 GB land behind it; the grid is rebuilt, 60 steps are timed with the library's
HIP events, everything is freed.  Diagnostic only (profiles/r06/README.md section 6).
    python3 tools/region_probe.py [sizes in GB, comma separated] [repeats]"""
import gc
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from otmb_amd import synthetic_device

sizes = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,16,64,128,192,240,0").split(",")]
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = []
for rep in range(repeats):
    for gb in sizes:
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info()
        n = int(gb * (1 << 30))
        if n > free - (12 << 30):
            out.append({"dummy_gb": gb, "skipped": f"only {free >> 30} GB free"})
            continue
        dummy = torch.empty(n, dtype=torch.uint8, device=dev) if n else None
        dg = synthetic_device.make_device_grid("access1deg", dev)
        asm = synthetic_device.assembler_for(dg)
        for _ in range(5):
            asm.step_async(dg.umo, dg.vmo, dg.fill)
        asm.finish()
        asm.ctx.timing_enable(True)
        for _ in range(60):
            asm.step_async(dg.umo, dg.vmo, dg.fill)
        asm.finish()
        k = asm.ctx.timing_collect()
        asm.ctx.timing_enable(False)
        rec = {"rep": rep, "dummy_gb": gb, "fill_ms": round(k["tm_kernel<fill>"][0] / k["tm_kernel<fill>"][1], 5),
               "facefluxes_ms": round(k["facefluxes_kernel"][0] / k["facefluxes_kernel"][1], 5),
               "lowest_address_gb": round(min(t.data_ptr() for t in (asm.phi[0], asm.out["T"][2])) / (1 << 30), 2)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del asm, dg, dummy
        gc.collect()
