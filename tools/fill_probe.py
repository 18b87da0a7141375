"""Yardstick: what plain streams reach on this part -- torch fill_ (write only) and copy_ (read + write) of 61 MB .. 2 GB buffers
(r05: fill 6.7-7.0 TB/s, copy 5.1-5.6 TB/s beyond the Infinity Cache); bench.py measures the same each run (roofline.calibration)."""
import torch, time
dev = torch.device("cuda", 0)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (61, 245, 490, 1960):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    bufs = [torch.empty(n, device=dev) for _ in range(4)] if mb < 1000 else [a]
    i = [0]
    def fill():
        bufs[i[0] % len(bufs)].fill_(1.0); i[0] += 1
    def copy():
        b.copy_(a)
    uf, uc = t(fill), t(copy)
    print("%5d MB: fill %.1f us = %.2f TB/s | copy %.1f us = %.2f TB/s (read + write)" % (mb, uf, mb / uf, uc, 2 * mb / uc), flush=True)
