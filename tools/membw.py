#!/usr/bin/env python3
"""Practical HBM rates on this box (torch kernels): fill (write only), copy (1R+1W), sum (read only)."""
import torch, time
n = 1 << 28  # 2 GiB of float64
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
GB = n * 8 / 1e9
print(f"fill  {GB / t(lambda: a.fill_(1.0)):8.1f} GB/s (write)")
print(f"copy  {2 * GB / t(lambda: b.copy_(a)):8.1f} GB/s (read+write)")
print(f"sum   {GB / t(lambda: a.sum()):8.1f} GB/s (read)")
