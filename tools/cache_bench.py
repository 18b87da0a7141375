#!/usr/bin/env python3
"""Developer micro-benchmark: the batched cache tier alone (bench.py's configs[2] section without the side lines)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

torch.cuda.set_device(0)
ev = bench.make_tables(bench.KAGGLE_LN, 36, seed=0, device="cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
# argv[2] = number of timed (unseen) batches; with many of them and argv[3] = 0 (no replay settle) a rocprofv3 kernel
# trace of this script averages over batches that run AT capacity with fresh keys, not over replayed warm-up batches
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
settle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.35
r = bench.cache_tier_section(ev, bench.KAGGLE_LN, 36, B, torch.device("cuda"), steps=steps, batch1=False, settle_s=settle)
print(json.dumps(r))
