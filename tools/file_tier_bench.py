#!/usr/bin/env python3
"""Developer micro-benchmark: the batched cache in front of a FILE-backed miss tier in staged mode (every table served by
the host reader pool: new rows gathered by host threads into a pinned buffer, one copy per batch) against the zero-copy
host-memory tier (tools/host_tier_bench.py).  Tables are written to /dev/shm (page-cache speed: the reader pool's own cost).
usage: python tools/file_tier_bench.py [batches] [pinned_budget_GiB]"""
import json
import os
import shutil
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda")
ln, d, B = bench.KAGGLE_LN, 36, 16384
T = len(ln)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
budget = int(float(sys.argv[2]) * (1 << 30)) if len(sys.argv) > 2 else 0
root = "/dev/shm/evs_file_tier_bench"
os.makedirs(root, exist_ok=True)
try:
    ev = bench.make_tables(ln, d, seed=0, device=dev)
    paths = []
    for k, t in enumerate(ev.raw):
        p = os.path.join(root, "ev-table-%d.bin" % (k + 1))
        t.cpu().numpy().tofile(p)
        paths.append(p)
    del ev
    torch.cuda.empty_cache()
    tier = E.FileTier(paths, 4 * d, pinned_budget_bytes=budget)
    cap = int(0.10 * sum(ln))
    c = E.GpuCache("evlfu", cap, T, d, 32, "python", dev)
    c.set_file_backing(tier)
    batches = bench.make_batches(ln, B, 60 + steps, seed=3, device=dev, dist="zipf", alpha=0.75)
    rows = [b[1].t().contiguous().to(torch.int32) for b in batches]
    x = torch.rand((B, d), device=dev)
    F = T + 1
    out = torch.empty((B, d + F * (F - 1) // 2), device=dev)
    hit = torch.empty((B, T), dtype=torch.uint8, device=dev)
    for i in range(60):
        c.lookup_interact(rows[i], x, out=out, hit=hit)
    torch.cuda.synchronize()
    s0 = c.batch_stats()
    t0 = time.perf_counter()
    for i in range(steps):
        c.lookup_interact(rows[60 + i], x, out=out, hit=hit)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = c.batch_stats()
    print(json.dumps({"ms_per_batch": dt / steps * 1e3, "value": T * B * steps / dt, "hit_rate": (s1["n_hits"] - s0["n_hits"]) / (T * B * steps),
                      "registered_tables": int(sum(tier.registered)), "staged_rows": c.staged_rows()}))
finally:
    shutil.rmtree(root, ignore_errors=True)
