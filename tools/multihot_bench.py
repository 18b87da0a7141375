#!/usr/bin/env python3
"""Developer micro-benchmark: genuinely multi-hot bags (the reference's random generator: up to --num-indices-per-lookup
indices per bag, dlrm_data_pytorch.py:1024-1065) through the fused launch and the two-call path, Kaggle-shaped tables.
usage: python tools/multihot_bench.py [B] [max_bag]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
max_bag = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ln, d = bench.KAGGLE_LN, 36
T = len(ln)
ev = bench.make_tables(ln, d, seed=0, device=dev)
rs = np.random.RandomState(1)
batches = []
for _ in range(4):
    lo, li = [], []
    for n in ln:
        sizes = np.maximum(1, np.round(rs.rand(B) * min(n, max_bag))).astype(np.int64)
        off = np.zeros(B, np.int64)
        off[1:] = np.cumsum(sizes)[:-1]
        lo.append(torch.from_numpy(off).to(dev))
        li.append(torch.from_numpy(rs.randint(0, n, size=int(sizes.sum()))).to(dev))
    batches.append((lo, li))
x = torch.rand(B, d, device=dev)
nnz = sum(int(t.numel()) for t in batches[0][1])


def timeit(fn, iters=100):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


f = timeit(lambda i: E.apply_emb_interact(x, batches[i % 4][0], batches[i % 4][1], ev))
g = timeit(lambda i: E.apply_emb(batches[i % 4][0], batches[i % 4][1], ev, None, lazy=False))
t = timeit(lambda i: E.interact_features(x, E.apply_emb(batches[i % 4][0], batches[i % 4][1], ev, None, lazy=False)))
bytes_f = nnz * (4 * d + 8) + B * (T * 8 + 4 * d + 4 * (d + 351))
print("B=%d max_bag=%d: %d lookups per batch (%.1f per bag); fused %.1f us = %.2f G lookups/s, %.0f GB/s algorithmic; gather alone %.1f us; two-call %.1f us"
      % (B, max_bag, nnz, nnz / (B * T), f, nnz / f / 1e3, bytes_f / f / 1e3, g, t))
