#!/usr/bin/env python3
"""What a plain device-to-device copy reaches on this GPU (read + write bytes per second): the practical ceiling that the
streaming stages (pyramid, blur) are priced against in DESIGN.md beside the 8 TB/s of the data sheet."""
import time

import torch

for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(mb, "MB copy: read+write GB/s", round(2 * n / dt / 1e9, 1))
