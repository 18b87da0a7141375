#!/usr/bin/env python3
"""Developer probe (timing only): the reduced-precision fused launch over tables whose rows start every S bytes instead of
back to back.  Needs a library built with -DEVS_X_STRIDE=S (tools/variants.sh ...@evs_fused_rfq) -- with the shipped library the
rows are read at the tight stride and the figure is the usual one.  usage: stride_probe.py bits S [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import evstore_dlrm_amd as E
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kbench import timeit, settle

bits, S = int(sys.argv[1]), int(sys.argv[2])
d, T = 36, 26
row = d * bits // 8
g = torch.Generator(device="cuda").manual_seed(0)
ws = []
for n in bench.KAGGLE_LN:
    m = (n + row - 1) // row * row          # (n S bytes must reshape into rows of `row` bytes)
    ws.append(torch.randint(0, 15 if bits == 4 else 256, (m * S,), device="cuda", generator=g, dtype=torch.uint8).reshape(-1, row))
ev = E.EVTables(ws, d, bits)
for B in [int(a) for a in sys.argv[3:]] or [16384, 65536]:
    batches = bench.make_batches(bench.KAGGLE_LN, B, 8, 1, "cuda", "uniform")
    x = torch.rand(B, d, device="cuda")
    fn = lambda i: E.apply_emb_interact(x, batches[i % 8][0], batches[i % 8][1], ev, one_index_per_bag=True)
    settle(fn)
    print("u%d stride %d B=%d: %.1f us" % (bits, S, B, timeit(fn, 200)), flush=True)
