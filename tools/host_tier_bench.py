#!/usr/bin/env python3
"""Developer micro-benchmark: the batched cache in front of tables in PINNED HOST memory (the reference's C3 / mmap miss
path, bench.py's cache_tier.host_miss_tier line alone).  usage: python tools/host_tier_bench.py [batches]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda")
ln, d, B = bench.KAGGLE_LN, 36, 16384
T = len(ln)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ev = bench.make_tables(ln, d, seed=0, device=dev)
host = [t.cpu().pin_memory() for t in ev.raw]
cap = int(0.10 * sum(ln))
c = E.GpuCache("evlfu", cap, T, d, 32, "python", dev)
c.set_backing(host)
batches = bench.make_batches(ln, B, 60 + steps, seed=3, device=dev, dist="zipf", alpha=0.75)
rows = [b[1].t().contiguous().to(torch.int32) for b in batches]
x = torch.rand((B, d), device=dev)
F = T + 1
out = torch.empty((B, d + F * (F - 1) // 2), device=dev)
hit = torch.empty((B, T), dtype=torch.uint8, device=dev)
for i in range(60):
    c.lookup_interact(rows[i], x, out=out, hit=hit)
torch.cuda.synchronize()
s0 = c.batch_stats()
t0 = time.perf_counter()
for i in range(steps):
    c.lookup_interact(rows[60 + i], x, out=out, hit=hit)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s1 = c.batch_stats()
print(json.dumps({"ms_per_batch": dt / steps * 1e3, "value": T * B * steps / dt, "hit_rate": (s1["n_hits"] - s0["n_hits"]) / (T * B * steps),
                  "resident": s1["size"]}))
