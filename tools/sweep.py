#!/usr/bin/env python3
"""Measurement sweep over the configurations SURVEY.md 8(d) lists (one MI355X):
  cfg 2  Kaggle shape, all tables in HBM: B in {1,128,2048,16384,65536}, d in {36,16,64}, fp32 + u16/u8/u4
  cfg 3  EvLFU C1 at 10 % of the rows: batched path at B in {2048,16384}, batch-1 exact path
  cfg 4  Terabyte-shaped tables (MLPerf DLRM cardinalities capped at 40 M rows: external / synthetic), d=64 and 128
  cfg 5  C1 (u8) + C2 (u4) two-tier exact path, batch-1 and B=2048 replay
Prints a markdown report (committed as profiles/r01_sweep.md).  usage: python tools/sweep.py [--quick]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402
from evstore_dlrm_amd import gpu_cache  # noqa: E402

# MLPerf DLRM (Criteo Terabyte) cardinalities with --max-ind-range=40000000 (bench/run_and_time.sh:17);
# not in the reference tree: external, used as a synthetic shape only
TERABYTE_LN = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
               4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
HBM_PEAK = 8000.0


def per_batch_us(fn, n_batches, iters):
    """p50 / p95 / mean per-batch latency with one HIP event pair per batch (inputs resident)."""
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters):
        ev[i][0].record()
        fn(i)
        ev[i][1].record()
    torch.cuda.synchronize()
    lat = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return float(np.percentile(lat, 50)), float(np.percentile(lat, 95)), e0.elapsed_time(e1) * 1e3 / iters


def fused_rows(ln, d, bits, batches_B, iters, label, codes="random"):
    ev = bench.make_tables(ln, d, bits=bits, codes=codes)
    T = len(ln)
    P = (T + 1) * T // 2
    out = []
    for B in batches_B:
        bs = bench.make_batches(ln, B, 8, 1, "cuda", "uniform")
        x = torch.rand(B, d, device="cuda")
        R = torch.empty((B, d + P), device="cuda")
        row_b = d * bits // 8
        p50, p95, mean = per_batch_us(lambda i: E.apply_emb_interact(x, bs[i % 8][0], bs[i % 8][1], ev, out=R, one_index_per_bag=True), 8, iters)
        g50, _, gmean = per_batch_us(lambda i: E.apply_emb_interact(x, bs[i % 8][0], bs[i % 8][1], ev, out=R), 8, iters)
        alg = B * (T * (row_b + 8) + 4 * d + 4 * (d + P))
        out.append("| %s | %d | %d | %d | %.1f | %.1f | %.2f | %.0f | %.1f%% | %.1f | %.2f |" % (
            label, d, bits, B, p50, p95, T * B / mean / 1e3, alg / mean / 1e3, alg / mean / 1e3 / HBM_PEAK * 100, g50, T * B / gmean / 1e3))
        if bits == 32 and B >= 128:
            t50, _, tmean = per_batch_us(lambda i: E.interact_features(x, E.apply_emb(bs[i % 8][0], bs[i % 8][1], ev, None, lazy=False)), 8, max(20, iters // 4))
            out[-1] += " %.1f | %.2f |" % (t50, T * B / tmean / 1e3)
        else:
            out[-1] += " – | – |"
    del ev
    torch.cuda.empty_cache()
    return out


def cache_rows(ln, d, iters):
    """One cache, filled to capacity with 60 batches of 16 384 (Zipf 0.75), then timed at B = 2048 and 16 384."""
    dev = torch.device("cuda")
    ev = bench.make_tables(ln, d)
    T = len(ln)
    cap = int(0.10 * sum(ln))
    cache = E.GpuCache("evlfu", cap, T, d, 32, "python", dev)
    cache.set_backing(ev)
    F = T + 1
    warm = bench.make_batches(ln, 16384, 60, seed=3, device=dev, dist="zipf", alpha=0.75)
    x = torch.rand((16384, d), device=dev)
    R = torch.empty((16384, d + F * (F - 1) // 2), device=dev)
    hit = torch.empty((16384, T), dtype=torch.uint8, device=dev)
    for b in warm:
        cache.lookup_interact(b[1].t().contiguous().to(torch.int32), x, out=R, hit=hit)
    del warm
    out = []
    for B in (2048, 16384):
        bs = bench.make_batches(ln, B, iters, seed=11 + B, device=dev, dist="zipf", alpha=0.75)
        rows = [b[1].t().contiguous().to(torch.int32) for b in bs]
        s0 = cache.batch_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in rows:
            cache.lookup_interact(r, x[:B], out=R[:B], hit=hit[:B])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s1 = cache.batch_stats()
        out.append("| batched EvLFU C1, %d entries = 10 %% of the rows (full), Zipf 0.75 | %d | %.1f | %.2f | %.3f | %d |" % (
            s1["size"], B, dt / iters * 1e6, T * B * iters / dt / 1e9, (s1["n_hits"] - s0["n_hits"]) / (T * B * iters),
            s1["n_evict"] - s0["n_evict"]))
    # the same cache in front of tables that live in pinned HOST memory (the reference's C3 / mmap miss path):
    # misses cross the bus inside the consumer kernel and the fill kernel; no cache = every row crosses it
    try:
        host = [t.cpu().pin_memory() for t in ev.raw]
        ch = E.GpuCache("evlfu", cap, T, d, 32, "python", dev)
        ch.set_backing(host)
        warm = bench.make_batches(ln, 16384, 60, seed=3, device=dev, dist="zipf", alpha=0.75)
        for b in warm:
            ch.lookup_interact(b[1].t().contiguous().to(torch.int32), x, out=R, hit=hit)
        del warm
        bs = bench.make_batches(ln, 16384, iters, seed=77, device=dev, dist="zipf", alpha=0.75)
        rows = [b[1].t().contiguous().to(torch.int32) for b in bs]
        s0 = ch.batch_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in rows:
            ch.lookup_interact(r, x, out=R, hit=hit)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s1 = ch.batch_stats()
        out.append("| the same cache, tables (miss tier) in pinned host memory | 16384 | %.1f | %.2f | %.3f | %d |" % (
            dt / iters * 1e6, T * 16384 * iters / dt / 1e9, (s1["n_hits"] - s0["n_hits"]) / (T * 16384 * iters), s1["n_evict"] - s0["n_evict"]))
        evh = E.EVTables(host, d, 32, device=dev)   # host-resident tables straight into the fused kernel (no cache)
        idx = [b[1] for b in bs[:4]]
        off = torch.arange(16384, device=dev).repeat(T, 1)
        for i in range(2):
            E.apply_emb_interact(x, off, idx[i], evh, out=R, one_index_per_bag=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(8):
            E.apply_emb_interact(x, off, idx[i % 4], evh, out=R, one_index_per_bag=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append("| no cache, every row read from pinned host memory by the fused kernel | 16384 | %.1f | %.3f | – | – |" % (
            dt / 8 * 1e6, T * 16384 * 8 / dt / 1e9))
        del ch, host
    except Exception as ex:   # pinning 4.9 GB can fail on a small box
        out.append("| host-memory miss tier | – | failed: %r | | | |" % (ex,))
    return out, ev


def exact_rows(ev, ln, d, n_req):
    """batch-1 exact paths: C1 fp32 (cfg 3) and C1 u8 + C2 u4 (cfg 5); ids and rows in pinned host buffers."""
    T = len(ln)
    dev = torch.device("cuda")
    bs = bench.make_batches(ln, 256, (n_req + 255) // 256 + 1, seed=5, device=dev, dist="zipf", alpha=0.75)
    reqs = torch.cat([b[1].t().contiguous().to(torch.int32) for b in bs])[:n_req].contiguous()
    host = reqs.cpu()
    rows = torch.empty((1, T), dtype=torch.int32).pin_memory()
    o = torch.empty((1, T, d), dtype=torch.float32).pin_memory()
    h = torch.empty((1, T), dtype=torch.uint8).pin_memory()
    out = []
    cap = 200000
    c = E.GpuCache("evlfu", cap, T, d, 32, "python", dev)
    c.set_backing(ev)
    lat = []
    for i in range(n_req):
        t0 = time.perf_counter()
        rows.copy_(host[i:i + 1])
        c.request(rows, out=o, hit=h)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e6)
    st = c.stats()
    out.append("| C1 fp32 exact (cfg 3), %d entries | 1 | %.1f | %.1f | %.3f |" % (cap, np.percentile(lat[200:], 50), np.percentile(lat[200:], 95), st["n_hits"] / (T * n_req)))
    t0 = time.perf_counter()
    c.request(reqs[:2048].contiguous())
    torch.cuda.synchronize()
    out.append("| C1 fp32 exact, 2048 requests replayed in one launch | 2048 | %.1f per request | – | – |" % ((time.perf_counter() - t0) * 1e6 / 2048))
    del c
    # the same stream through the HOST engine of the exact policy (evs_hostcache_*: what ev_lookup runs by default),
    # tables copied to host memory, one request per call
    from evstore_dlrm_amd import host_cache as HC
    tabs_h = [t.cpu().numpy() for t in ev.raw]
    hc = HC.HostCache("evlfu", cap, T, d, 32, "python").set_backing(tabs_h)
    hr = host.numpy()
    o_np, h_np = np.empty((1, T, d), np.float32), np.empty((1, T), np.uint8)
    lat = []
    for i in range(n_req):
        t0 = time.perf_counter()
        hc.request(hr[i:i + 1], out=o_np, hit=h_np)
        lat.append((time.perf_counter() - t0) * 1e6)
    sth = hc.stats()
    out.append("| C1 fp32 exact, HOST engine (evs_hostcache_request through the Python handle), %d entries | 1 | %.1f | %.1f | %.3f |" % (
        cap, np.percentile(lat[200:], 50), np.percentile(lat[200:], 95), sth["n_hits"] / (T * n_req)))
    t0 = time.perf_counter()
    hc.request(hr[:2048])
    out.append("| C1 fp32 exact, HOST engine, 2048 requests in one call | 2048 | %.2f per request | – | – |" % ((time.perf_counter() - t0) * 1e6 / 2048))
    del hc, tabs_h
    ev8, ev4 = bench.make_tables(ln, d, bits=8, seed=8), bench.make_tables(ln, d, bits=4, seed=4)
    c1 = E.GpuCache("evlfu", 9000, T, d, 8, "cpp", dev)      # 1 : 2 entries like "48-48-4" (evlfu_8.cpp:63-78), small enough
    c2 = E.GpuCache("evlfu", 18000, T, d, 4, "cpp", dev)     # that C1 fills and the odd/even routing to C2 starts
    c1.set_backing(ev8)
    c2.set_backing(ev4)
    tier = torch.empty((1, T), dtype=torch.uint8, device=dev)
    od = torch.empty((1, T, d), device=dev)
    rd = torch.empty((1, T), dtype=torch.int32, device=dev)
    lat = []
    n_c1 = n_c2 = 0
    for i in range(n_req):
        t0 = time.perf_counter()
        rd.copy_(host[i:i + 1], non_blocking=True)
        gpu_cache.request_c1c2(c1, c2, rd, out=od, tier=tier)
        o.copy_(od, non_blocking=True)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e6)
    t_all, _ = gpu_cache.request_c1c2(c1, c2, reqs[:2048].contiguous())
    n_c1, n_c2 = int((t_all == 1).sum()), int((t_all == 2).sum())
    out.append("| C1 u8 (9 000) + C2 u4 (18 000) exact (cfg 5) | 1 | %.1f | %.1f | C1 %.3f / C2 %.3f (of the next 2048 requests) |" % (
        np.percentile(lat[200:], 50), np.percentile(lat[200:], 95), n_c1 / (2048 * T), n_c2 / (2048 * T)))
    del c1, c2
    # the same two-tier request stream through the HOST engine (evs_hostcache_request_c1c2c3: what ev_lookup runs by default
    # with N_CACHING_LAYER = 2), encoded tables in host memory
    h1 = HC.HostCache("evlfu", 9000, T, d, 8, "cpp").set_backing([t.cpu().numpy() for t in ev8.raw])
    h2 = HC.HostCache("evlfu", 18000, T, d, 4, "cpp").set_backing([t.cpu().numpy() for t in ev4.raw])
    t_np = np.empty((1, T), np.uint8)
    lat = []
    for i in range(n_req):
        t0 = time.perf_counter()
        HC.request_c1c2c3(h1, h2, None, hr[i:i + 1], out=o_np, tier=t_np)
        lat.append((time.perf_counter() - t0) * 1e6)
    t_all, _ = HC.request_c1c2c3(h1, h2, None, hr[:2048])
    out.append("| C1 u8 (9 000) + C2 u4 (18 000) exact (cfg 5), HOST engine | 1 | %.1f | %.1f | C1 %.3f / C2 %.3f (of the next 2048 requests) |" % (
        np.percentile(lat[200:], 50), np.percentile(lat[200:], 95), float((t_all == 1).sum()) / (2048 * T), float((t_all == 2).sum()) / (2048 * T)))
    del h1, h2
    # cfg 5 as a throughput path: batched two-tier lookup (snapshot semantics), tiers sized 48 % : 48 % of a budget
    # of 2 % of the rows in fp32-row equivalents (u8 entries cost 1/4, u4 entries 1/8: evlfu_8.cpp:63-78)
    budget = int(0.02 * sum(ln))   # 2 % so that C1 fills inside the warm-up and the routing to C2 starts
    cap1, cap2 = int(0.48 * budget) * 4, int(0.48 * budget) * 8
    cap1, cap2 = min(cap1, sum(ln)), min(cap2, sum(ln))
    b1 = E.GpuCache("evlfu", cap1, T, d, 8, "cpp", dev)
    b2 = E.GpuCache("evlfu", cap2, T, d, 4, "cpp", dev)
    b1.set_backing(ev8)
    b2.set_backing(ev4)
    Bq = 16384
    outq = torch.empty((Bq, T, d), device=dev)
    tierq = torch.empty((Bq, T), dtype=torch.uint8, device=dev)
    warm = bench.make_batches(ln, Bq, 180, seed=21, device=dev, dist="zipf", alpha=0.75)   # both tiers full before the timed batches
    for b in warm:
        gpu_cache.lookup_batch_c1c2(b1, b2, b[1].t().contiguous().to(torch.int32), out=outq, tier=tierq)
    del warm
    bsq = bench.make_batches(ln, Bq, 40, seed=22, device=dev, dist="zipf", alpha=0.75)
    rq = [b[1].t().contiguous().to(torch.int32) for b in bsq]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n1 = n2 = 0
    for r in rq:
        gpu_cache.lookup_batch_c1c2(b1, b2, r, out=outq, tier=tierq)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n1, n2 = int((tierq == 1).sum()), int((tierq == 2).sum())
    s1, s2 = b1.batch_stats(), b2.batch_stats()
    out.append("| batched C1 u8 (%d entries, %d resident) + C2 u4 (%d entries, %d resident), rows decoded to fp32 (cfg 5) | %d | %.1f per batch = %.2f G lookups/s | – | C1 %.3f / C2 %.3f (last batch) |" % (
        cap1, s1["size"], cap2, s2["size"], Bq, dt / len(rq) * 1e6, T * Bq * len(rq) / dt / 1e9, n1 / (Bq * T), n2 / (Bq * T)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only-reduced", action="store_true", help="the reduced-precision rows alone")
    a = ap.parse_args()
    it = 50 if a.quick else 200
    print("# Round-4 sweep (one MI355X, synthetic uniform indices unless stated; HBM peak used: 8 000 GB/s)\n")
    print("`fused` = `apply_emb_interact` (one kernel; one index per bag declared); `offsets` = the same with `lS_o` read and")
    print("validated; `two-call` = `apply_emb(lazy=False)` then `interact_features`: two kernels, the pooled rows in HBM (with lazy pooling, the default, the pair runs as the fused launch).  Latencies are per batch, HIP events, inputs resident.")
    print("GB/s = algorithmic bytes (SURVEY 8(d): rows + indices + x read, R written) / mean batch time.\n")
    print("| shape | d | bits | B | fused p50 µs | p95 µs | G lookups/s | GB/s | of peak | offsets p50 µs | G lookups/s | two-call p50 µs | G lookups/s |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    Bs = [1, 128, 2048, 16384] if a.quick else [1, 128, 2048, 16384, 65536]
    if a.only_reduced:
        for bits in (16, 8, 4):
            for line in fused_rows(bench.KAGGLE_LN, 36, bits, [2048, 16384, 65536], it, "Kaggle, reduced precision, encoded tables", "encoded"):
                print(line, flush=True)
        for line in fused_rows(bench.KAGGLE_LN, 36, 16, [16384, 65536], it, "Kaggle, reduced precision, random codes"):
            print(line, flush=True)
        return
    for line in fused_rows(bench.KAGGLE_LN, 36, 32, Bs, it, "Kaggle (cfg 2)"):
        print(line, flush=True)
    for d in (16, 64):
        for line in fused_rows(bench.KAGGLE_LN, d, 32, [2048, 16384], it, "Kaggle"):
            print(line, flush=True)
    # reduced precision: tables ENCODED from the fp32 init by the reference's encoders (what reduce_precision.py produces:
    # no u16 code of the |x| > 0.65 tail), and uniformly random codes (0.8 % tail codes: the u16 worst case)
    for bits in (16, 8, 4):
        for line in fused_rows(bench.KAGGLE_LN, 36, bits, [2048, 16384] if a.quick else [2048, 16384, 65536], it, "Kaggle, reduced precision, encoded tables", "encoded"):
            print(line, flush=True)
    for line in fused_rows(bench.KAGGLE_LN, 36, 16, [16384] if a.quick else [16384, 65536], it, "Kaggle, reduced precision, random codes"):
        print(line, flush=True)
    for d in (64, 128):
        for line in fused_rows(TERABYTE_LN, d, 32, [2048, 16384], it, "Terabyte-shaped (cfg 4, 1 GPU)"):
            print(line, flush=True)
    # multi-hot bags (apply_emb alone and apply_emb + interact_features): the reference's benchmark shape first
    # (bench/dlrm_s_benchmark.sh:20-45), then other widths / bag lengths
    print("\n| long bags: tables x rows, d, indices per bag (fixed) | B | apply_emb µs | TB/s | of peak | G lookups/s | + interact_features µs |")
    print("|---|---|---|---|---|---|---|")
    for (B_, bag, d_, T_, rows_) in ([(2048, 100, 64, 8, 1000000), (2048, 100, 36, 8, 1000000)] if a.quick else
                                     [(2048, 100, 64, 8, 1000000), (2048, 100, 128, 8, 1000000), (2048, 100, 32, 8, 1000000), (2048, 100, 36, 8, 1000000),
                                      (2048, 100, 16, 8, 1000000), (2048, 38, 36, 26, 200000), (4096, 17, 36, 26, 200000)]):
        r = bench.long_bags_section(torch.device("cuda"), B=B_, bag=bag, d=d_, T=T_, rows=rows_)
        print("| %d x %d rows, d = %d, %d per bag | %d | %.1f | %.2f | %.1f%% | %.2f | %.1f |"
              % (T_, rows_, d_, bag, B_, r["apply_emb"]["ms_per_step"] * 1e3, r["apply_emb"]["achieved"] / 1e3, r["apply_emb"]["frac"] * 100,
                 r["apply_emb"]["value"] / 1e9, r["apply_emb_interact"]["ms_per_step"] * 1e3), flush=True)
    print("\n| cache tier | B | µs per batch | G lookups/s | hit rate | evictions in the timed batches |")
    print("|---|---|---|---|---|---|")
    rows, ev = cache_rows(bench.KAGGLE_LN, 36, 30)
    for line in rows:
        print(line, flush=True)
    print("\n| exact (reference batch-1 semantics), ids and rows on the host | B | p50 µs | p95 µs | hit rate |")
    print("|---|---|---|---|---|")
    for line in exact_rows(ev, bench.KAGGLE_LN, 36, 800 if a.quick else 1500):
        print(line, flush=True)


if __name__ == "__main__":
    main()
