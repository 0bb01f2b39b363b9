import os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/wavecap-sdr_amd']
import torch, wavehip
n = 1 << 28
ch = wavehip.PolyphaseChannelizer(10_000_000, 9765)
x = torch.view_as_complex(torch.randn(n, 2, device="cuda").mul_(0.5))
stats = torch.zeros((1024, 5), dtype=torch.float64, device="cuda")
for _ in range(5): ch.process_stats_device(x, stats)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ch.process_stats_device(x, stats)
torch.cuda.synchronize(); print("stats-only M=1024 2^28:", (time.perf_counter() - t0) / 10 * 1e3, "ms")
