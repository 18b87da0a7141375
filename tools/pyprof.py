"""Developer: host-side cost of the two-call plugin surface at a small batch (cProfile)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
ln = [min(n, 100000) for n in bench.KAGGLE_LN]
ev = bench.make_tables(ln, 36)
B = 128
bs = bench.make_batches(ln, B, 4, 1, "cuda", "uniform")
x = torch.rand(B, 36, device="cuda")
def step(i):
    ly = E.apply_emb(bs[i % 4][0], bs[i % 4][1], ev, None, lazy=False)
    return E.interact_features(x, ly)
for i in range(20): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(2000): step(i)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
