#!/usr/bin/env python3
"""Developer experiment: does the fused launch slow down because the tiny Kaggle tables (3..27 rows) put every
sample's request on the same few cache lines?  Same kernel, same bytes; only the row counts of the tiny tables change."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402
from tools.kbench import timeit  # noqa: E402

for name, ln in (("kaggle", bench.KAGGLE_LN),
                 ("tiny->2k rows", [max(n, 2000) for n in bench.KAGGLE_LN]),
                 ("small->200k rows", [max(n, 200000) for n in bench.KAGGLE_LN]),
                 ("all 1M rows", [1000000] * 26),
                 ("all 3 rows", [3] * 26)):
    ev = bench.make_tables(ln, 36)
    for B in (16384, 65536):
        batches = bench.make_batches(ln, B, 8, 1, "cuda", "uniform")
        x = torch.rand(B, 36, device="cuda")
        us = timeit(lambda i: E.apply_emb_interact(x, batches[i % 8][0], batches[i % 8][1], ev, one_index_per_bag=True), 300)
        print("%-18s B=%6d %7.1f us" % (name, B, us), flush=True)
    del ev
    torch.cuda.empty_cache()
