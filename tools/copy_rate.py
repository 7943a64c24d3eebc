#!/usr/bin/env python3
"""What a plain device copy reaches on this box (GB/s read + written), for the sizes of a 4096-proof phase: the ceiling
the streaming kernels are compared with next to the 8 TB/s datasheet peak."""
import torch, time
for mb in (168, 336, 1024):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.int64, device="cuda").random_()
    b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"copy {mb} MB -> {mb} MB: {us:.1f} us, {2 * mb * 1.048576 / us * 1e3:.0f} GB/s (read + write)")
    c = torch.empty(n, dtype=torch.int64, device="cuda")
    e0.record()
    for _ in range(50):
        s = a.sum()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"read-only reduction of {mb} MB: {us:.1f} us, {mb * 1.048576 / us * 1e3:.0f} GB/s")
