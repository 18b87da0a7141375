import sys, torch, time
sys.path.insert(0, '/root/repo')
import bench, evstore_dlrm_amd as E
dev = torch.device('cuda')
ev = bench.make_tables(bench.KAGGLE_LN, 36)
r = bench.cache_tier_section(ev, bench.KAGGLE_LN, 36, 16384, dev, steps=30, warmup=20)
print("ms/step %.4f hit %.3f" % (r["ms_per_step"], r["hit_rate"]))
