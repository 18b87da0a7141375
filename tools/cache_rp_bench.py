"""Developer micro-benchmark: a single REDUCED-PRECISION tier (the reference's one-layer evlfu_16 / _8 / _4 builds: MAIN_PRECISION 16 / 8 / 4,
N_CACHING_LAYER 1) through the batched lookup + interaction, at its 10 % capacity in steady state.  python tools/cache_rp_bench.py [bits ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import evstore_dlrm_amd as E

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
T, d, B = 26, 36, int(os.environ.get("B", 16384))
ev = bench.make_tables(bench.KAGGLE_LN, d, seed=0, device=dev)
rq = [b[1].t().contiguous().to(torch.int32) for b in bench.make_batches(bench.KAGGLE_LN, B, 60 + 200, seed=3, device=dev, dist="zipf", alpha=0.75)]
x = torch.rand((B, d), device=dev)
out = torch.empty((B, d + 351), device=dev)
hit = torch.empty((B, T), dtype=torch.uint8, device=dev)
for bits in [int(a) for a in sys.argv[1:]] or [8, 4, 16, 32]:
    evq = ev.encode(bits) if bits != 32 else ev
    c = E.GpuCache("evlfu", int(0.10 * sum(bench.KAGGLE_LN)), T, d, bits, "python", dev)
    c.set_backing(evq)
    for i in range(60):
        c.lookup_interact(rq[i], x, out=out, hit=hit)
    s0 = c.batch_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(200):
        c.lookup_interact(rq[60 + i], x, out=out, hit=hit)
    e1.record()
    torch.cuda.synchronize()
    s1 = c.batch_stats()
    print("u%-2d tier: %.1f us per batch, hit rate %.4f" % (bits, e0.elapsed_time(e1) / 200 * 1e3, (s1["n_hits"] - s0["n_hits"]) / (T * B * 200)), flush=True)
    del c, evq
