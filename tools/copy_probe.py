"""Yardstick: torch's own device-to-device copy at the bench's byte volumes (what a launch that only moves bytes reaches)."""
import torch, time
torch.cuda.set_device(0)
for mb in (32, 64, 128, 512):
    n = mb * 1024 * 1024 // 4
    xs = [torch.rand(n, device="cuda") for _ in range(6)]
    ys = [torch.empty(n, device="cuda") for _ in range(6)]
    for i in range(20): ys[i % 6].copy_(xs[i % 6])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(120): ys[i % 6].copy_(xs[i % 6])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 120 * 1e3
    print("copy %4d MB: %.1f us  read+write %.0f GB/s" % (mb, us, 2 * mb * 1.048576 / us * 1e3))
