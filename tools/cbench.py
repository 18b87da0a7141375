"""Developer micro-benchmark: the cache tier's batched lookups by policy (per-batch stream time), the target of tools/prof_cache.sh."""
import sys, torch, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
dev = torch.device('cuda')
ev = bench.make_tables(bench.KAGGLE_LN, 36)
for alpha in [float(a) for a in sys.argv[1:]] or [1.05]:
    r = bench.cache_tier_section(ev, bench.KAGGLE_LN, 36, 16384, dev, steps=100, warmup=60, alpha=alpha, batch1=False)
    print("alpha %.2f ms/step %.4f hit %.3f resident %d evictions %d" % (alpha, r["ms_per_step"], r["hit_rate"], r["resident_entries"], r["evictions"]))
