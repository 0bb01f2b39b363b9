#!/usr/bin/env python3
"""Read-only / write-only / copy device bandwidth with plain torch ops (yardsticks for the roofline discussion)."""
import torch
n = 1 << 30                      # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty(n, dtype=torch.float32, device="cuda")
def timed(fn, bytes_moved, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return bytes_moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
print("write-only (fill_)      %7.1f GB/s" % timed(lambda: b.fill_(1.5), 4 * n))
print("write-only (zero_)      %7.1f GB/s" % timed(lambda: b.zero_(), 4 * n))
print("read-only  (sum)        %7.1f GB/s" % timed(lambda: a.sum(), 4 * n))
print("copy 1:1   (copy_)      %7.1f GB/s" % timed(lambda: b.copy_(a), 8 * n))
print("scale 1:1  (mul out=)   %7.1f GB/s" % timed(lambda: torch.mul(a, 2.0, out=b), 8 * n))
h = n // 2
print("1 read : 2 writes       %7.1f GB/s" % timed(lambda: b.view(2, h).copy_(a[:h].expand(2, -1)), 12 * h))
