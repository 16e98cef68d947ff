#!/usr/bin/env python3
"""Practical HBM ceiling of this GPU for the two access shapes of the two-pass engine:
a streaming copy (read one buffer, write another) and an in-place read-modify-write, both
over 4 GiB with 16-byte accesses (torch elementwise kernels).  Prints GB/s (read + write bytes).
  python tools/hbm_ceiling.py"""
import time

import torch

dev = torch.device("cuda", 0)
N = 1 << 30   # int32 elements = 4 GiB
a = torch.zeros(N, dtype=torch.int32, device=dev)
b = torch.empty_like(a)


def bench(fn, bytes_moved, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return bytes_moved * reps / (time.perf_counter() - t0) / 1e9


print(f"copy 4 GiB -> 4 GiB      : {bench(lambda: b.copy_(a), 2 * 4 * N):8.1f} GB/s")
print(f"in-place x += 1 on 4 GiB : {bench(lambda: a.add_(1), 2 * 4 * N):8.1f} GB/s")
