"""Developer probe: the Prefetcher loop piece by piece."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
from evstore_dlrm_amd import inference_loop as IL
dev = torch.device("cuda")
ln, d, B, T = bench.KAGGLE_LN, 36, 16384, 26
ev = bench.make_tables(ln, d)
g = torch.Generator().manual_seed(5)
host = []
for _ in range(8):
    li = torch.stack([torch.randint(0, n, (B,), generator=g) for n in ln])
    host.append((torch.rand(B, 13, generator=g), torch.arange(B).repeat(T, 1).contiguous(), li))
x = torch.rand(B, d, device=dev); out = torch.empty(B, 36 + 351, device=dev)
def run(tag, n, body, wire=torch.int64):
    pk = IL.PackedPinnedBatches(host, 8, wire)
    pf = IL.Prefetcher(pk, dev)
    for X, lo, li in pf: body(X, lo, li)
    pk.count = n
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for X, lo, li in pf:
        body(X, lo, li)
    torch.cuda.synchronize(); print("%-40s %.3f ms per batch" % (tag, (time.perf_counter() - t0) / n * 1e3))
run("prefetch only", 100, lambda X, lo, li: None)
run("prefetch + fused launch", 100, lambda X, lo, li: E.apply_emb_interact(x, lo, li, ev, None, out=out))
run("prefetch + fused launch (declared)", 100, lambda X, lo, li: E.apply_emb_interact(x, lo, li, ev, None, out=out, one_index_per_bag=True))
run("int32 wire + fused launch", 100, lambda X, lo, li: E.apply_emb_interact(x, lo, li, ev, None, out=out), torch.int32)
pf = IL.Prefetcher(IL.PackedPinnedBatches(host, 100), dev)
t0 = time.perf_counter(); [pf._issue(i) for i in range(2)]; torch.cuda.synchronize(); print("2 issues %.3f ms" % ((time.perf_counter() - t0) * 1e3))
