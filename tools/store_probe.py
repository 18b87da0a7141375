"""Yardstick for launches whose bytes are mostly fp32 OUTPUT (apply_emb alone on reduced-precision tables at B = 65 536:
245 MB of stores behind 61 MB of row reads): what torch's own fill and copy kernels reach at that size on this part --
outputs that fit the 256 MB Infinity Cache and outputs that do not.  python tools/store_probe.py"""
import torch
torch.cuda.set_device(0)


def timed(f, n=60):
    for _ in range(10):
        f(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        f(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for mb in (61, 123, 245, 490):
    n = mb * 1000 * 1000 // 4
    ys = [torch.empty(n, device="cuda") for _ in range(4)]
    xs = [torch.rand(n // 4, device="cuda") for _ in range(4)]          # a quarter of the bytes read, as for u8 rows
    src = [torch.rand(n, device="cuda") for _ in range(4)]
    t_fill = timed(lambda i: ys[i % 4].fill_(1.0))
    t_copy = timed(lambda i: ys[i % 4].copy_(src[i % 4]))
    t_exp = timed(lambda i: torch.mul(xs[i % 4].view(-1, 1).expand(-1, 4), 2.0, out=ys[i % 4].view(-1, 4)))   # read 1, write 4
    print("%4d MB out: fill %.1f us = %.2f TB/s written | copy %.1f us = %.2f TB/s each way | read 1/4 + write %.1f us = %.2f TB/s written"
          % (mb, t_fill, mb / t_fill, t_copy, mb / t_copy, t_exp, mb / t_exp))
