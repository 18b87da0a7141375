#!/usr/bin/env python3
"""Developer probe: what the brackets of a short timed region cost on this stack (idle synchronise, event record / query,
launch-to-completion of one tiny kernel, stream.synchronize vs device synchronize)."""
import time
import torch

torch.cuda.set_device(0)
y = torch.zeros(16, device="cuda")
for _ in range(200):
    y.add_(1)
torch.cuda.synchronize()


def med(f, n=200):
    v = []
    for _ in range(n):
        t = time.perf_counter(); f(); v.append((time.perf_counter() - t) * 1e6)
    v.sort()
    return v[len(v) // 2], v[int(len(v) * 0.95)]


print("idle torch.cuda.synchronize()            p50 %.1f us  p95 %.1f" % med(torch.cuda.synchronize))
st = torch.cuda.current_stream()
print("idle current_stream().synchronize()      p50 %.1f us  p95 %.1f" % med(st.synchronize))
e = torch.cuda.Event(enable_timing=True)
e.record(); torch.cuda.synchronize()
print("event.query() on a completed event       p50 %.1f us  p95 %.1f" % med(e.query))
print("event.record()                           p50 %.1f us  p95 %.1f" % med(lambda: e.record()))
torch.cuda.synchronize()


def one_blocking():
    y.add_(1); torch.cuda.synchronize()


def one_stream():
    y.add_(1); st.synchronize()


def one_poll():
    y.add_(1); e.record()
    while not e.query():
        pass


print("tiny kernel + torch.cuda.synchronize()   p50 %.1f us  p95 %.1f" % med(one_blocking))
print("tiny kernel + stream.synchronize()       p50 %.1f us  p95 %.1f" % med(one_stream))
print("tiny kernel + record + poll query()      p50 %.1f us  p95 %.1f" % med(one_poll))
