#!/usr/bin/env python3
"""Developer probe: where a request of the reference's batch-1 loop (dlrm_wrap + apply_emb_evstore, use_gpu=True) spends its
time -- the three host-to-device copies of dlrm_wrap (reference harness code) vs this package's forward."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from evstore_dlrm_amd import evstore_ops  # noqa: E402
from evstore_dlrm_amd.cache_algo import EvLFU_C1  # noqa: E402
from evstore_dlrm_amd.emb_storage import storage_manager as sm  # noqa: E402

dev = torch.device("cuda")
d, T, n = 36, 26, 3000
ev = bench.make_tables(bench.KAGGLE_LN, d, seed=0, device="cuda")
sm.use_device_tables([t.cpu() for t in ev.raw], 32, storage=sm.EmbStorage.DUMMY)
EvLFU_C1.init(200000, engine="host")
evstore_ops.cache_algo = "evlfu"
b1s = bench.make_batches(bench.KAGGLE_LN, 256, (n + 255) // 256, seed=13, device=dev, dist="zipf", alpha=1.05)
rows = torch.cat([b[1].t().contiguous() for b in b1s])[:n].cpu()
X = torch.zeros(1, 13)
lS_o = torch.zeros((T, 1), dtype=torch.int64)
ld = [(X, lS_o, rows[i].reshape(T, 1)) for i in range(n)]
for rep in range(2):
    t_copy, t_fwd, t_tot = [], [], []
    for Xh, oh, ih in ld:
        t0 = time.perf_counter()
        i_d = ih.to(dev); o_d = oh.to(dev); x_d = Xh.to(dev)
        t1 = time.perf_counter()
        ly = evstore_ops.apply_emb_evstore(o_d, i_d, None, None, use_gpu=True, use_emb_cache=True)
        t2 = time.perf_counter()
        t_copy.append(t1 - t0); t_fwd.append(t2 - t1); t_tot.append(t2 - t0)
print("p50 per request: dlrm_wrap's three .to(device) %.1f us, apply_emb_evstore(use_gpu=True) %.1f us, both %.1f us"
      % (np.median(t_copy) * 1e6, np.median(t_fwd) * 1e6, np.median(t_tot) * 1e6))
