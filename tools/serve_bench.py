#!/usr/bin/env python3
"""Developer micro-benchmark: the fused call launched (apply_emb_interact + polled end event) against the resident dispatcher
(InteractServer): p50 of ONE batch posted and waited for, and per-batch time of batches posted back to back, by batch size."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import evstore_dlrm_amd as E

dev = torch.device("cuda")
ln, d, T = bench.KAGGLE_LN, 36, 26
ev = bench.make_tables(ln, d)
P = (T + 1) * T // 2
srv = E.InteractServer(ev, idle_us=int(os.environ.get("IDLE_US", "200")), n_blocks=int(os.environ.get("N_BLOCKS", "0")))
for B in [int(a) for a in sys.argv[1:]] or [1, 128, 2048, 16384]:
    bs = bench.make_batches(ln, B, 32, seed=5 + B, device=dev)
    x = torch.rand((B, d), device=dev)
    outs = [torch.empty((B, d + P), device=dev) for _ in range(64)]
    want = E.apply_emb_interact(x, bs[0][0], bs[0][1], ev)
    torch.cuda.synchronize()
    got = srv(x, bs[0][0], bs[0][1])
    assert torch.equal(got, want)
    done = torch.cuda.Event()
    done.record()
    lat_l, lat_s = [], []
    for i in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        E.apply_emb_interact(x, bs[i % 32][0], bs[i % 32][1], ev, None, out=outs[0])
        done.record()
        while not done.query():
            pass
        lat_l.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    for i in range(300):
        t0 = time.perf_counter()
        srv(x, bs[i % 32][0], bs[i % 32][1], out=outs[0])
        lat_s.append((time.perf_counter() - t0) * 1e6)
    # back to back: n batches posted (the ring holds 64), the last one waited for
    n = 2000
    for rep in range(2):
        t0 = time.perf_counter()
        tk = None
        for i in range(n):
            tk = srv.post(x, bs[i % 32][0], bs[i % 32][1], out=outs[i % 64])[0]
        srv.wait(tk)
        per_s = (time.perf_counter() - t0) / n * 1e6
    srv.stop()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record(); b_.record(); torch.cuda.synchronize()
    a_.record()
    for i in range(n):
        E.apply_emb_interact(x, bs[i % 32][0], bs[i % 32][1], ev, None, out=outs[i % 64])
    b_.record(); torch.cuda.synchronize()
    per_l = a_.elapsed_time(b_) / n * 1e3
    print("B=%6d  one batch, waited for: launched p50 %.2f us (p95 %.2f), resident p50 %.2f us (p95 %.2f)   back to back: launched %.2f us / batch, resident %.2f"
          % (B, np.percentile(lat_l, 50), np.percentile(lat_l, 95), np.percentile(lat_s, 50), np.percentile(lat_s, 95), per_l, per_s), flush=True)
srv.close()
