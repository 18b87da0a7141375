#!/usr/bin/env python3
"""Developer probe (GPU box): why / whether K batches per call over a stream pair reach the two-streams rate.
Variants, B = 16 384, per-batch stream time and host time: single stream; two torch streams, no events; fork / join per
group of K through torch events (both side streams; one side stream + the caller's); the library's multi call."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402


def main():
    B, d, T = 16384, 36, 26
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ev = bench.make_tables(bench.KAGGLE_LN, d)
    bs = bench.make_batches(bench.KAGGLE_LN, B, 64, 1, "cuda", "uniform")
    x = torch.rand(B, d, device="cuda")
    Rs = [torch.empty((B, d + (T + 1) * T // 2), device="cuda") for _ in range(K)]
    cur = torch.cuda.current_stream()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    evf = torch.cuda.Event()
    evj = [torch.cuda.Event(), torch.cuda.Event()]

    def one(i, k):
        E.apply_emb_interact(x, None, bs[i % 64][1], ev, out=Rs[k], one_index_per_bag=True)

    def single(g):
        for k in range(K):
            one(g * K + k, k)

    def two_noev(g):
        for k in range(K):
            with torch.cuda.stream(s0 if k % 2 == 0 else s1):
                one(g * K + k, k)

    def forkjoin2(g):
        evf.record(cur)
        s0.wait_event(evf)
        s1.wait_event(evf)
        for k in range(K):
            with torch.cuda.stream(s0 if k % 2 == 0 else s1):
                one(g * K + k, k)
        evj[0].record(s0)
        evj[1].record(s1)
        cur.wait_event(evj[0])
        cur.wait_event(evj[1])

    def forkjoin1(g):   # even batches on the caller's stream, odd ones on ONE side stream
        evf.record(cur)
        s1.wait_event(evf)
        for k in range(K):
            if k % 2 == 0:
                one(g * K + k, k)
            else:
                with torch.cuda.stream(s1):
                    one(g * K + k, k)
        evj[1].record(s1)
        cur.wait_event(evj[1])

    lis = [[bs[(g * K + k) % 64][1] for k in range(K)] for g in range(64 // K)]
    xs = [x] * K

    def multi(g):
        E.apply_emb_interact_multi(xs, None, lis[g % len(lis)], ev, outs=Rs, one_index_per_bag=True)

    def run(name, fn, n=150):
        for g in range(20):
            fn(g)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        a.record()
        for g in range(n):
            fn(g)
        host = time.perf_counter() - t
        b.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t
        print("%-46s stream %6.2f us/batch   host %6.2f us/batch   wall %6.2f us/batch" % (
            name, a.elapsed_time(b) * 1e3 / (n * K), host * 1e6 / (n * K), wall * 1e6 / (n * K)), flush=True)

    for _ in range(2):
        run("single stream", single)
        run("two torch streams, no events", two_noev)
        run("fork/join per %d batches, two side streams" % K, forkjoin2)
        run("fork/join per %d batches, caller + one side stream" % K, forkjoin1)
        run("library multi call (K = %d)" % K, multi)


if __name__ == "__main__":
    main()
