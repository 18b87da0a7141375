#!/usr/bin/env python3
"""Host-side cost of one call into the fused launch (GPU box): how many microseconds of CPU time the caller spends per
apply_emb_interact, through the C++ extension and through ctypes, next to the floor of the stack (an empty-stream torch op,
a bare event record).  N back-to-back calls without synchronising (the queue absorbs them), wall-clock / N.
usage: python tools/call_overhead.py [B]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402
from evstore_dlrm_amd import _ext  # noqa: E402


def per_call(fn, n=3000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / n * 1e6


def gpu_per_call(fn, n=3000):
    for _ in range(200):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    d, T = 36, 26
    ev = bench.make_tables(bench.KAGGLE_LN, d)
    off, idx = bench.make_batches(bench.KAGGLE_LN, B, 1, 1, "cuda", "uniform")[0]
    x = torch.rand(B, d, device="cuda")
    R = torch.empty((B, d + (T + 1) * T // 2), device="cuda")
    y = torch.zeros(16, device="cuda")
    e = torch.cuda.Event(enable_timing=True)
    rows = []

    def line(name, fn):
        h, g = per_call(fn), gpu_per_call(fn)
        rows.append((name, h, g))
        print("%-58s host %6.2f us/call   stream %6.2f us/call" % (name, h, g), flush=True)

    line("torch: y.add_(1) on 16 floats", lambda: y.add_(1))
    line("torch: event.record()", lambda: e.record())
    X = _ext.ext()
    assert X is not None
    xt = ev.ext_tables()
    line("extension, raw: X.apply_emb_interact(out=R, declared)", lambda: X.apply_emb_interact(xt, x, None, idx, False, R, True, False))
    line("extension: E.apply_emb_interact(out=R, declared)", lambda: E.apply_emb_interact(x, off, idx, ev, out=R, one_index_per_bag=True))
    line("extension: E.apply_emb_interact(out=R, lS_o given)", lambda: E.apply_emb_interact(x, off, idx, ev, out=R))
    line("extension: E.apply_emb_interact(allocating R)", lambda: E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True))
    line("extension: interact_features(x, apply_emb(...))", lambda: E.interact_features(x, E.apply_emb(off, idx, ev, None, lazy=False)))
    saved = (_ext._mod, _ext._tried)
    _ext._mod, _ext._tried = None, True
    ev._xt = None
    try:
        line("ctypes: E.apply_emb_interact(out=R, declared)", lambda: E.apply_emb_interact(x, off, idx, ev, out=R, one_index_per_bag=True))
        line("ctypes: interact_features(x, apply_emb(...))", lambda: E.interact_features(x, E.apply_emb(off, idx, ev, None, lazy=False)))
    finally:
        _ext._mod, _ext._tried = saved
    print("\n| call (B = %d) | host us/call | stream us/call |\n|---|---|---|" % B)
    for r in rows:
        print("| %s | %.2f | %.2f |" % r)


if __name__ == "__main__":
    main()
