#!/usr/bin/env python3
"""Developer: the two-tier (configs[4] pair: u8 C1 + u4 C2, the reference's 48-48-4 split of 2 % of the rows) batched lookup +
interaction ALONE -- 180 batches fill both tiers, then argv[1] (default 200) unseen batches at capacity, per-batch stream time
by HIP events.  The profile target of tools/prof_r06.sh: every dispatch of the two kernels behind the fill is a steady-state one.
argv[2] = 3: with the alt-key tier (three tiers)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import evstore_dlrm_amd as E
from evstore_dlrm_amd import gpu_cache

dev = torch.device("cuda")
ln, d, T, B = bench.KAGGLE_LN, 36, 26, 16384
n_timed = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tiers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = bench.make_tables(ln, d)
ev8, ev4 = ev.encode(8), ev.encode(4)
budget = int(0.02 * sum(ln))
c1 = E.GpuCache("evlfu", int(0.48 * budget) * 4, T, d, 8, "cpp", dev)
c2 = E.GpuCache("evlfu", int(0.48 * budget) * 8, T, d, 4, "cpp", dev)
c1.set_backing(ev8); c2.set_backing(ev4)
c3 = None
if tiers == 3:
    alt = [torch.from_numpy(((np.arange(n, dtype=np.int64) % min(n, 4096)) * 100 + (t + 1)).astype(np.uint32).view(np.int32)).to(dev)
           for t, n in enumerate(ln)]
    c3 = E.GpuAltKeyTier(int(0.04 * budget) * 8 + 64, alt, dev)
tier = torch.empty((B, T), dtype=torch.uint8, device=dev)
x = torch.rand((B, d), device=dev)
P = (T + 1) * T // 2
R = torch.empty((B, d + P), device=dev)
n_fill = 180
bs = bench.make_batches(ln, B, n_fill + n_timed, seed=21, device=dev, dist="zipf", alpha=0.75)
rq = [b[1].t().contiguous().to(torch.int32) for b in bs]
del bs


def step(r):
    if c3 is None:
        gpu_cache.lookup_interact_c1c2(c1, c2, r, x, tier=tier, fused=True)
    else:
        gpu_cache.lookup_interact_c1c2c3(c1, c2, c3, r, x, tier=tier)


for r in rq[:n_fill]:
    step(r)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); b.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
a.record()
for r in rq[n_fill:]:
    step(r)
b.record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d-tier batched + interaction: %.2f us per batch by events, %.2f wall (%d unseen batches at capacity); C1 %d C2 %d resident"
      % (tiers, a.elapsed_time(b) / n_timed * 1e3, dt / n_timed * 1e6, n_timed, c1.batch_stats()["size"], c2.batch_stats()["size"]))
