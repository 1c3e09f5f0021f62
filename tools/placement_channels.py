#!/usr/bin/env python3
"""does the spread of the fill pass's time between output sets (the "allocation lottery", 5.85 ... 6.57 ms at 0.25 degree inside one
process) follow an IMBALANCE OF THE HBM CHANNELS behind the sets' physical pages?  Run under
    rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_RDREQ --kernel-trace --output-format json -d <dir> -- python3 tools/placement_channels.py quarterdeg 4 3
One assembler, NSETS output sets, the fill pass launched REPS times into every set in turn (set s = dispatches [s * REPS, (s + 1) * REPS) of
tm_kernel<0>), each launch also timed with the library's HIP events; the launch order and the event times go to <out>.json, the per-channel counter
values come from rocprofv3's JSON (tools/tcc_channels.py --per-dispatch)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from otmb_amd import synthetic_device

wl = sys.argv[1] if len(sys.argv) > 1 else "quarterdeg"
nsets = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "placement_channels.json")
dev = torch.device("cuda", 0)
dg = synthetic_device.make_device_grid(wl, dev, seed=20260501, rho="array")
asm = synthetic_device.assembler_for(dg, 0)
phi = asm.facefluxes(dg.umo, dg.vmo, dg.fill)
asm.count_in_ff = False  # (every launch below counts for itself: the same tm_count_kernel + scan in front of every fill)
asm.transportmatrix_onepass(phi)
sets = [asm.out]
rng = np.random.default_rng(7)
for s in range(1, nsets):
    spacer = torch.empty(int(rng.integers(3, 400)) * (1 << 20) + 4096 * int(rng.integers(0, 255)), dtype=torch.uint8, device=dev)
    sets.append(asm.new_output_set())
    del spacer
order, times = [], []
for k, out in enumerate(sets):
    for _ in range(reps):
        asm.ctx.timing_enable(True)
        asm.transportmatrix_onepass(phi, sync=False, out=out)
        asm.result()
        t = asm.ctx.timing_collect()
        asm.ctx.timing_enable(False)
        order.append(k)
        times.append(round(t["tm_kernel<fill>"][0] / t["tm_kernel<fill>"][1], 5))
addr = [{m: [hex(s[m][q].data_ptr()) for q in range(3)] for m in s} for s in sets]
json.dump({"workload": wl, "sets": nsets, "reps": reps, "dispatch_set": [0] + order, "event_ms": [None] + times, "addr": addr, "N": asm.N, "nnz": asm.nnz,
           "note": "dispatch 0 is the warm-up launch into set 0; event_ms: the library's HIP events around the fill kernel, under the counters"},
          open(out_path, "w"))
print(out_path)
