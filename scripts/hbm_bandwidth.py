#!/usr/bin/env python3
"""What plain streaming kernels reach on this GPU (torch copy / sum / fill over 512 MB): the
yardstick for the measured HBM rate of the iteration kernel.   python scripts/hbm_bandwidth.py"""
import torch, time
x = torch.empty(512 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")  # 512 MB
y = torch.empty_like(x)
x.normal_()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
dt = t(lambda: y.copy_(x)); print(f"copy 512 MB: {2*x.numel()*4/dt/1e12:.2f} TB/s (read+write)")
dt = t(lambda: x.sum()); print(f"read-only sum 512 MB: {x.numel()*4/dt/1e12:.2f} TB/s")
dt = t(lambda: y.fill_(1.0)); print(f"write-only fill 512 MB: {x.numel()*4/dt/1e12:.2f} TB/s")
a = x[: 200 * 1024 * 1024 // 4]; b = y[: 200 * 1024 * 1024 // 4]
dt = t(lambda: b.copy_(a)); print(f"copy 200 MB: {2*a.numel()*4/dt/1e12:.2f} TB/s (read+write)")
