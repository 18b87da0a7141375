#!/usr/bin/env python3
"""Developer micro-benchmark: per-kernel timings of the hot path on one GPU.
usage: python tools/kbench.py [--batch 16384] [--layouts tile,table] [--iters 200]"""
import argparse
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def settle(fn, seconds=0.3):
    """Run fn until the clocks have ramped (the first timed loop of a process otherwise reads 10-20 % slow)."""
    import time
    t0 = time.time()
    i = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn(i)
            i += 1
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[16384])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--dim", type=int, default=36)
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--n-batches", type=int, default=8, help="distinct batches cycled (bench.py: 64 = index data far beyond the Infinity Cache)")
    ap.add_argument("--fused-only", action="store_true")
    ap.add_argument("--bits", type=int, default=32, help="table precision 32|16|8|4 (reduced: fused timings only)")
    ap.add_argument("--codes", default="random", help="reduced precision: random codes | encoded (the fp32 init through the reference's encoders)")
    a = ap.parse_args()
    d = a.dim
    ev = bench.make_tables(bench.KAGGLE_LN, d, bits=a.bits, codes=a.codes)
    T = 26
    for B in a.batch:
        batches = bench.make_batches(bench.KAGGLE_LN, B, a.n_batches, 1, "cuda", a.dist)
        x = torch.rand(B, d, device="cuda")
        if a.bits != 32:
            row = d * a.bits // 8
            settle(lambda i: E.apply_emb_interact(x, batches[i % len(batches)][0], batches[i % len(batches)][1], ev))
            f_us = timeit(lambda i: E.apply_emb_interact(x, batches[i % len(batches)][0], batches[i % len(batches)][1], ev), a.iters)
            f1_us = timeit(lambda i: E.apply_emb_interact(x, batches[i % len(batches)][0], batches[i % len(batches)][1], ev, one_index_per_bag=True), a.iters)
            g_us = timeit(lambda i: E.apply_emb(batches[i % len(batches)][0], batches[i % len(batches)][1], ev, None, lazy=False), a.iters)
            fb = B * (26 * (row + 8) + 4 * d + 4 * (d + 351)) / 1e3
            print("B=%6d u%d fused one-index/bag %7.1f us (%5.0f GB/s algorithmic, %.2f G lookups/s) | offsets %7.1f us | gather only %7.1f us"
                  % (B, a.bits, f1_us, fb / f1_us, 26 * B / f1_us / 1e3, f_us, g_us), flush=True)
            continue
        tile = torch.empty(B, T + 1, d, device="cuda")
        g_tile = timeit(lambda i: E.apply_emb(batches[i % len(batches)][0], batches[i % len(batches)][1], ev, None, out=tile), a.iters)
        ly_tile = E.apply_emb(batches[0][0], batches[0][1], ev, None, out=tile)
        i_tile = timeit(lambda i: E.interact_features(x, ly_tile), a.iters)
        g_tab = timeit(lambda i: E.apply_emb(batches[i % len(batches)][0], batches[i % len(batches)][1], ev, None, lazy=False), a.iters)
        ly_tab = E.apply_emb(batches[0][0], batches[0][1], ev, None, lazy=False)
        i_tab = timeit(lambda i: E.interact_features(x, ly_tab), a.iters)
        f_us = timeit(lambda i: E.apply_emb_interact(x, batches[i % len(batches)][0], batches[i % len(batches)][1], ev), a.iters)
        f1_us = timeit(lambda i: E.apply_emb_interact(x, batches[i % len(batches)][0], batches[i % len(batches)][1], ev, one_index_per_bag=True), a.iters)
        print("B=%6d  fused, one index per bag   %7.1f us (%.2f G lookups/s)" % (B, f1_us, 26 * B / f1_us / 1e3))
        fb = B * (26 * (4 * d + 16) + 4 * d + 4 * (d + 351)) / 1e3
        print("B=%6d  fused gather+interact %7.1f us (%5.0f GB/s algorithmic, %.2f G lookups/s)" % (B, f_us, fb / f_us, 26 * B / f_us / 1e3))
        if a.fused_only:
            continue
        gb = T * B * (8 * d + 16) / 1e3  # KB... bytes/1e3
        ib = B * (4 * 27 * d + 4 * (d + 351)) / 1e3
        print("B=%6d  gather tile %7.1f us (%5.0f GB/s)  table %7.1f us (%5.0f GB/s) | interact tile %7.1f us (%5.0f GB/s)  table %7.1f us (%5.0f GB/s)"
              % (B, g_tile, gb / g_tile, g_tab, gb / g_tab, i_tile, ib / i_tile, i_tab, ib / i_tab), flush=True)


if __name__ == "__main__":
    main()
