#!/usr/bin/env python3
"""Developer tool: the batched cache tier batch by batch.  Runs W warm-up + N unseen Zipf batches at capacity, reads the
policy counters after every batch (synchronising: this is a diagnosis run, not a timing run) and prints per batch:
evictions, flushes, free entries, tombstones, rebuild, device time of the batch (HIP events).  Under
`rocprofv3 --kernel-trace` the per-dispatch CSV can be joined with this table (tools/cache_trace_join.py)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W = 60
dev = torch.device("cuda")
torch.cuda.set_device(0)
ln, d = bench.KAGGLE_LN, 36
T = len(ln)
ev = bench.make_tables(ln, d, seed=0, device="cuda")
cap = int(0.10 * sum(ln))
cache = E.GpuCache("evlfu", cap, T, d, 32, "python", dev)
cache.set_backing(ev)
batches = bench.make_batches(ln, B, W + N, seed=3, device=dev, dist="zipf", alpha=0.75)
rows = [b[1].t().contiguous().to(torch.int32) for b in batches]
x = torch.rand((B, d), device=dev)
F = T + 1
out = torch.empty((B, d + F * (F - 1) // 2), device=dev)
hit = torch.empty((B, T), dtype=torch.uint8, device=dev)
prev = cache.batch_stats() if False else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("batch   us  evict  flush   free    tomb   hits  perfect_bucket")
for i in range(W + N):
    e0.record()
    cache.lookup_interact(rows[i], x, out=out, hit=hit)
    e1.record()
    torch.cuda.synchronize()
    s = cache.batch_stats()
    if prev is not None and i >= W - 5:
        print("%5d %5.0f %6d %6d %6d %7d %6d %7d" % (i, e0.elapsed_time(e1) * 1e3, s["n_evict"] - prev["n_evict"],
                                                   s["n_flush"] - prev["n_flush"], s["n_free"], s["n_tomb"],
                                                   s["n_hits"] - prev["n_hits"], s["hist"][T]))
    prev = s
print(json.dumps({"cap": cap, "hist": prev["hist"]}))
