#!/usr/bin/env python3
"""A DLRM inference loop on the drop-in ops, end to end: bottom MLP (torch) -> apply_emb + interact_features (this package's
HIP path, the reference's two calls as written: dlrm_s_pytorch.py:588-605 sequential_forward) -> top MLP (torch), fed by
the packed-pinned loader through the copy-stream prefetcher, timed like the reference's loop (inference_loop.inference).

    python examples/dlrm_inference_demo.py            # Criteo-Kaggle cardinalities, d = 36 (4.9 GB of tables)
    python examples/dlrm_inference_demo.py --small    # 26 small tables (seconds)

Synthetic weights and indices (there is no dataset here); the arithmetic is the model's."""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import evstore_dlrm_amd as evs                      # noqa: E402
from evstore_dlrm_amd import inference_loop as IL   # noqa: E402

KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652, 2173, 4,
          7046547, 18, 15, 286181, 105, 142572]


def mlp(sizes, sigmoid_last=False):
    layers = []
    for i in range(len(sizes) - 1):
        layers += [nn.Linear(sizes[i], sizes[i + 1]), nn.Sigmoid() if (sigmoid_last and i == len(sizes) - 2) else nn.ReLU()]
    return nn.Sequential(*layers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--requests", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ln = [min(n, 5000) for n in KAGGLE] if a.small else KAGGLE
    d, T, B = 36, len(ln), (512 if a.small else a.batch)
    g = torch.Generator(device=dev).manual_seed(0)
    emb_l = [torch.empty((n, d), device=dev).uniform_(-(1.0 / n) ** 0.5, (1.0 / n) ** 0.5, generator=g) for n in ln]
    ev_tables = evs.EVTables.from_fp32(emb_l)                       # the tables stay in HBM in the .bin byte layout
    bot = mlp([13, 512, 256, 64, d]).to(dev)
    F = T + 1
    top = mlp([d + F * (F - 1) // 2, 512, 256, 1], sigmoid_last=True).to(dev)

    @torch.no_grad()
    def forward(X, lS_o, lS_i):
        x = bot(X)                                                  # apply_mlp(dense_x, self.bot_l)
        ly = evs.apply_emb(lS_o, lS_i, ev_tables, None)             # a real list; the gather is deferred ...
        z = evs.interact_features(x, ly, "dot", False)              # ... and runs fused with the interaction here
        return top(z)                                               # apply_mlp(z, self.top_l)

    host = []
    for _ in range(8):
        lS_i = torch.stack([torch.randint(0, n, (B,)) for n in ln])
        host.append((torch.rand(B, 13), torch.arange(B).repeat(T, 1).contiguous(), lS_i))
    ld = IL.PackedPinnedBatches(host, a.requests)                   # every batch ONE pinned block
    for X, lo, li in IL.Prefetcher(IL.PackedPinnedBatches(host, 8), dev):
        forward(X, lo, li)                                          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = None
    for X, lo, li in IL.Prefetcher(ld, dev):                        # batch i + 1 crosses the bus under batch i's launches
        p = forward(X, lo, li)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert p.shape == (B, 1) and bool(torch.isfinite(p).all())
    print("DLRM forward, %d tables x d=%d, B=%d: %.3f ms per batch, %.2f G lookups/s, H2D included (click probability of sample 0: %.4f)"
          % (T, d, B, dt / a.requests * 1e3, T * B * a.requests / dt / 1e9, float(p[0])))


if __name__ == "__main__":
    main()
