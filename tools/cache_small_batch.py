#!/usr/bin/env python3
"""Developer micro-benchmark: the cache tier's lookup_interact at small batches (per-call latency, synchronised per call, and
stream time per call in a back-to-back loop).  usage: python tools/cache_small_batch.py [B ...]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

d, T = 36, 26
ev = bench.make_tables(bench.KAGGLE_LN, d, seed=0, device="cuda")
cap = int(0.10 * sum(bench.KAGGLE_LN))
for B in [int(a) for a in sys.argv[1:]] or [128, 512, 2048, 8192]:
    cache = E.GpuCache("evlfu", cap, T, d, 32, "python", torch.device("cuda"))
    cache.set_backing(ev)
    rows = [b[1].t().contiguous().to(torch.int32) for b in bench.make_batches(bench.KAGGLE_LN, B, 64, seed=3, device=torch.device("cuda"), dist="zipf", alpha=0.75)]
    x = torch.rand(B, d, device="cuda")
    out = torch.empty(B, d + 351, device="cuda")
    for i in range(300):
        cache.lookup_interact(rows[i % 64], x, out=out)
    torch.cuda.synchronize()
    lat = []
    for i in range(400):
        t0 = time.perf_counter()
        cache.lookup_interact(rows[i % 64], x, out=out)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(400):
        cache.lookup_interact(rows[i % 64], x, out=out)
    e1.record()
    torch.cuda.synchronize()
    print("B=%5d: p50 %.1f us per synchronised call, %.1f us per call back to back"
          % (B, float(np.median(lat)) * 1e6, e0.elapsed_time(e1) / 400 * 1e3), flush=True)
