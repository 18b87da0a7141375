"""Developer: per-kernel profile target for the batched two-tier lookup (run under tools/prof_any.sh)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
from evstore_dlrm_amd import gpu_cache
dev = torch.device("cuda")
ln, d, T = bench.KAGGLE_LN, 36, 26
ev = bench.make_tables(ln, d)
ev8, ev4 = ev.encode(8), ev.encode(4)
budget = int(0.02 * sum(ln))
c1 = E.GpuCache("evlfu", int(0.48 * budget) * 4, T, d, 8, "cpp", dev)
c2 = E.GpuCache("evlfu", int(0.48 * budget) * 8, T, d, 4, "cpp", dev)
c1.set_backing(ev8); c2.set_backing(ev4)
B = 16384
out = torch.empty((B, T, d), device=dev); tier = torch.empty((B, T), dtype=torch.uint8, device=dev)
bs = bench.make_batches(ln, B, 200, seed=21, device=dev, dist="zipf", alpha=0.75)   # 180 to fill both tiers, 20 timed
rq = [b[1].t().contiguous().to(torch.int32) for b in bs]
ONLY3 = os.environ.get("C2BENCH_ONLY3") == "1"   # profile target: fill, then the three-tier lookups alone
for r in rq[:180]:
    gpu_cache.lookup_batch_c1c2(c1, c2, r, out=out, tier=tier)
torch.cuda.synchronize(); t0 = time.perf_counter()
for r in rq[180:]:
    gpu_cache.lookup_batch_c1c2(c1, c2, r, out=out, tier=tier)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("two-tier batched, fp32 rows out: %.1f us per batch, %.2f G lookups/s, C1 %d C2 %d resident" % (dt / 20 * 1e6, T * B * 20 / dt / 1e9, c1.batch_stats()["size"], c2.batch_stats()["size"]))
# the same lookup feeding the interaction: rows materialised then dense interaction, vs decoded inside the consumer
x = torch.rand((B, d), device=dev)
bs2 = bench.make_batches(ln, B, 60, seed=22, device=dev, dist="zipf", alpha=0.75)
rq2 = [b[1].t().contiguous().to(torch.int32) for b in bs2]
for fused in (() if ONLY3 else (False, True)):
    for r in rq2[:10]:
        gpu_cache.lookup_interact_c1c2(c1, c2, r, x, out=out, tier=tier, fused=fused)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in rq2[10:30] if not fused else rq2[30:50]:
        gpu_cache.lookup_interact_c1c2(c1, c2, r, x, out=out, tier=tier, fused=fused)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("two-tier batched + interaction (%s): %.1f us per batch, %.2f G lookups/s"
          % ("mixed-codec consumer" if fused else "fp32 rows, then dense interaction", dt / 20 * 1e6, T * B * 20 / dt / 1e9))
# three tiers (configs[4]): the same two tiers + the alt-key tier C3 (alt key of (t, r): row r % 4096 of the same table,
# so alt rows are hot rows and likely resident)
import numpy as np
alt = [torch.from_numpy(((np.arange(n, dtype=np.int64) % min(n, 4096)) * 100 + (t + 1)).astype(np.uint32).view(np.int32)).to(dev)
       for t, n in enumerate(ln)]
c3 = E.GpuAltKeyTier(int(0.04 * budget) * 8 + 64, alt, dev)
bs3 = bench.make_batches(ln, B, 70, seed=23, device=dev, dist="zipf", alpha=0.75)
rq3 = [b[1].t().contiguous().to(torch.int32) for b in bs3]
for r in rq3[:30]:
    gpu_cache.lookup_interact_c1c2c3(c1, c2, c3, r, x, tier=tier)
torch.cuda.synchronize(); t0 = time.perf_counter()
n3 = 0
for r in rq3[30:70]:
    gpu_cache.lookup_interact_c1c2c3(c1, c2, c3, r, x, tier=tier)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
m3, st3 = c3.batch_dump()
print("three-tier batched + interaction (mixed-codec consumer): %.1f us per batch, %.2f G lookups/s; C3 %d members of %d, %d alt hits served"
      % (dt / 40 * 1e6, T * B * 40 / dt / 1e9, st3["members"], st3["capacity"], st3["n_hit"]))
