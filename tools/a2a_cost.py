#!/usr/bin/env python3
"""Developer probe: what one RCCL all_to_all_single costs through torch.distributed on this stack (world size 1: a
self-exchange, i.e. the call path without any link time) -- host time per call and stream time per call, async_op on/off."""
import os, sys, time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29581")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 16384 * 5 * 36
send = torch.rand(n, device=dev)
recv = torch.empty(n, device=dev)
y = torch.zeros(16, device=dev)


def run(fn, iters=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / iters * 1e6
    torch.cuda.synchronize()
    return host, e0.elapsed_time(e1) / iters * 1e3


def a_async():
    w = dist.all_to_all_single(recv, send, [n], [n], async_op=True)
    w.wait()


def a_sync():
    dist.all_to_all_single(recv, send, [n], [n], async_op=False)


def a_nosplit():
    dist.all_to_all_single(recv, send, async_op=False)


for name, fn in (("y.add_(1)  (one launch, for scale)", lambda: y.add_(1)), ("recv.copy_(send)  (the same bytes as a device copy)", lambda: recv.copy_(send)),
                 ("all_to_all_single(async_op=True) + wait()", a_async), ("all_to_all_single(async_op=False)", a_sync),
                 ("all_to_all_single(no split lists, async_op=False)", a_nosplit)):
    h, s = run(fn)
    print("%-58s host %6.1f us/call   stream %6.1f us/call" % (name, h, s))
dist.destroy_process_group()
