#!/usr/bin/env python3
"""Developer micro-benchmark: long bags.  Default = the reference's own benchmark shape (bench/dlrm_s_benchmark.sh:20-45:
8 tables x 1 M rows x d = 64, 100 indices per bag, fixed, mb 2 048): apply_emb alone and apply_emb + interact_features.
usage: python tools/longbag_bench.py [B] [bag] [d] [T] [rows]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bag = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = int(sys.argv[3]) if len(sys.argv) > 3 else 64
T = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = int(sys.argv[5]) if len(sys.argv) > 5 else 1000000
r = bench.long_bags_section(dev, B=B, bag=bag, d=d, T=T, rows=rows)
print("B=%d bag=%d d=%d T=%d rows=%d: apply_emb %.1f us = %.2f TB/s (%.3f of peak), %.2f G lookups/s; + interact_features %.1f us"
      % (B, bag, d, T, rows, r["apply_emb"]["ms_per_step"] * 1e3, r["apply_emb"]["achieved"] / 1e3, r["apply_emb"]["frac"],
         r["apply_emb"]["value"] / 1e9, r["apply_emb_interact"]["ms_per_step"] * 1e3))
