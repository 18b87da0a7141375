"""Developer micro-benchmark: evs_cache_lookup_batch (hit flags + the (B, T, d) fp32 rows out, no interaction) of a single tier at its
10 % capacity in steady state, by table precision.  python tools/cache_rows_bench.py [bits ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import evstore_dlrm_amd as E

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
T, d, B = 26, 36, 16384
ev = bench.make_tables(bench.KAGGLE_LN, d, seed=0, device=dev)
rq = [b[1].t().contiguous().to(torch.int32) for b in bench.make_batches(bench.KAGGLE_LN, B, 60 + 200, seed=3, device=dev, dist="zipf", alpha=0.75)]
out = torch.empty((B, T, d), device=dev)
hit = torch.empty((B, T), dtype=torch.uint8, device=dev)
for bits in [int(a) for a in sys.argv[1:]] or [32, 8]:
    evq = ev.encode(bits) if bits != 32 else ev
    c = E.GpuCache("evlfu", int(0.10 * sum(bench.KAGGLE_LN)), T, d, bits, "python", dev)
    c.set_backing(evq)
    for i in range(60):
        c.lookup_batch(rq[i], out=out, hit=hit)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(200):
        c.lookup_batch(rq[60 + i], out=out, hit=hit)
    e1.record()
    torch.cuda.synchronize()
    print("u%-2d tier, rows out: %.1f us per batch" % (bits, e0.elapsed_time(e1) / 200 * 1e3), flush=True)
    del c, evq
