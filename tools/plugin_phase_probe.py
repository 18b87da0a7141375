#!/usr/bin/env python3
"""Developer probe: where the time of one apply_emb_evstore(use_gpu=True) request goes with the GPU engine (the resident
server): the loop's pieces timed one by one over 2 000 requests (medians, us)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import evstore_dlrm_amd as E
E.configure_runtime()
from evstore_dlrm_amd import evstore_ops
from evstore_dlrm_amd import inference_loop as IL
from evstore_dlrm_amd.cache_algo import EvLFU_C1
from evstore_dlrm_amd.emb_storage import storage_manager as sm

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
T, d = 26, 36
ev = bench.make_tables(bench.KAGGLE_LN, d, seed=0, device=dev)
sm.use_device_tables(ev, 32)
EvLFU_C1.init(200000, engine="gpu")
evstore_ops.cache_algo = "evlfu"
b1s = bench.make_batches(bench.KAGGLE_LN, 256, 8, seed=13, device=dev, dist="zipf", alpha=1.05)
rows = torch.cat([b[1].t().contiguous() for b in b1s])[:2000].cpu()
lS_o = torch.zeros((T, 1), dtype=torch.int64)
m = EvLFU_C1._m
ph = {k: [] for k in ("to_dev", "cpu", "ids", "serve", "clone", "slices", "whole")}
for rep in range(2):
    for k in ph:
        ph[k].clear()
    for i in range(len(rows)):
        t0 = time.perf_counter()
        li = rows[i].reshape(T, 1).to(dev)
        t1 = time.perf_counter()
        lc = li.cpu().data
        t2 = time.perf_counter()
        ids = [int(s[0]) for s in lc.numpy()]
        t3 = time.perf_counter()
        hit, r = m._run(ids, -1, True)
        t4 = time.perf_counter()
        blk = r.detach().clone()
        t5 = time.perf_counter()
        from evstore_dlrm_amd import _ext
        X = _ext.ext()
        ly = X.slices(blk.unsqueeze(1), True) if X is not None else list(blk.unsqueeze(1).unbind(0))
        t6 = time.perf_counter()
        for k, a, b in (("to_dev", t0, t1), ("cpu", t1, t2), ("ids", t2, t3), ("serve", t3, t4), ("clone", t4, t5), ("slices", t5, t6), ("whole", t0, t6)):
            ph[k].append((b - a) * 1e6)
print({k: round(float(np.median(v)), 1) for k, v in ph.items()})
ld = [(torch.zeros(1, 13), lS_o, rows[i].reshape(T, 1)) for i in range(len(rows))]
fw = lambda X, o, i: evstore_ops.apply_emb_evstore(o, i, None, None, use_gpu=True, use_emb_cache=True)
IL.inference(ld, fw, True, dev)
st = IL.inference(ld, fw, True, dev)
print("through the loop: p50 %.1f us" % (IL.percentile_ms(st, 50) * 1e3))
