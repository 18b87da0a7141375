#!/usr/bin/env python3
"""Developer micro-benchmark: first top-MLP layer fused behind the interaction vs the unfused chain
(fused gather + interaction kernel, then torch's fp32 linear + ReLU = rocBLAS / hipBLASLt)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ev = bench.make_tables(bench.KAGGLE_LN, 36)
for B in (2048, 16384):
    batches = bench.make_batches(bench.KAGGLE_LN, B, 8, 1, "cuda")
    x = torch.rand(B, 36, device="cuda")
    for n1 in (512, 1024):
        W = torch.randn(n1, 387, device="cuda") * 0.1
        b = torch.randn(n1, device="cuda") * 0.1
        R = torch.empty(B, 387, device="cuda")
        f = timeit(lambda i: E.apply_emb_interact_mlp1(x, batches[i % 8][0], batches[i % 8][1], ev, W, b), 200)

        def unfused(i):
            E.apply_emb_interact(x, batches[i % 8][0], batches[i % 8][1], ev, None, out=R, one_index_per_bag=True)
            return torch.relu_(torch.addmm(b, R, W.t()))
        u = timeit(unfused, 200)
        g = timeit(lambda i: torch.relu_(torch.addmm(b, R, W.t())), 200)
        print("B=%6d n1=%4d  fused %7.1f us | unfused %7.1f us (of which torch addmm+relu %7.1f us) | %.1f TFLOP/s fused"
              % (B, n1, f, u, g, 2 * B * 387 * n1 / f / 1e6), flush=True)
