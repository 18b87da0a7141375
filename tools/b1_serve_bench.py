#!/usr/bin/env python3
"""Developer micro-benchmark: batch-1 latency of the GPU engine -- one launch + synchronise per request (evs_cache_request)
against the resident mailbox server (evs_cache_serve_*), same Zipf stream as bench.py's batch1_exact."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import bench, evstore_dlrm_amd as E
dev = torch.device("cuda")
ln, d, T = bench.KAGGLE_LN, 36, 26
ev = bench.make_tables(ln, d)
n1, n_skip, cap1 = 3000, 1000, 200000
b1s = bench.make_batches(ln, 256, (n1 + 255) // 256, seed=13, device=dev, dist="zipf", alpha=1.05)
req1 = torch.cat([b[1].t().contiguous().to(torch.int32) for b in b1s])[:n1].contiguous().cpu().numpy()
for mode in ("launch", "serve", "launch", "serve"):
    c = E.GpuCache("evlfu", cap1, T, d, 32, "python", dev); c.set_backing(ev)
    lat, hits = [], 0
    if mode == "serve":
        c.serve_start(n_slots=4, idle_us=int(os.environ.get("IDLE_US", "200")))
        for i in range(n1):
            t1 = time.perf_counter()
            h, rows = c.serve_request(req1[i])
            lat.append((time.perf_counter() - t1) * 1e6)
            if i >= n_skip: hits += int(h.sum())
        c.serve_stop()
    else:
        pr = torch.empty((1, T), dtype=torch.int32).pin_memory(); po = torch.empty((1, T, d), dtype=torch.float32).pin_memory(); ph = torch.empty((1, T), dtype=torch.uint8).pin_memory()
        for i in range(n1):
            t1 = time.perf_counter()
            pr[0] = torch.from_numpy(req1[i])
            c.request(pr, out=po, hit=ph)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e6)
            if i >= n_skip: hits += int(ph.sum())
    l = np.array(lat[n_skip:])
    print("%-6s p50 %.1f us  p95 %.1f us  mean %.1f us  hits %d" % (mode, np.percentile(l, 50), np.percentile(l, 95), l.mean(), hits))
    del c
