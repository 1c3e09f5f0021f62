"""Does this box's PCIe link carry uploads and downloads at the same time?  Pinned host buffers, two streams.
Prints GB/s of H2D alone, D2H alone, and both together (each direction's own rate while the other runs)."""
import time

import torch

n = 1 << 30
h_up = torch.empty(n, dtype=torch.uint8).pin_memory()
h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(n, dtype=torch.uint8, device="cuda")
d_dn = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, dn, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1):
                d_up.copy_(h_up, non_blocking=True)
        if dn:
            with torch.cuda.stream(s2):
                h_dn.copy_(d_dn, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for _ in range(2):
    run(True, True, 1)
tu, td, tb = run(True, False), run(False, True), run(True, True)
print(f"H2D alone {n / tu / 1e9:.1f} GB/s, D2H alone {n / td / 1e9:.1f} GB/s, both at once: {n / tb / 1e9:.1f} GB/s each way "
      f"({2 * n / tb / 1e9:.1f} GB/s aggregate; sequential would take {(tu + td) * 1e3:.1f} ms, together {tb * 1e3:.1f} ms)")
