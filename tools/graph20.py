"""Developer: K bench steps eager vs as ONE captured HIP graph, both bracketed by torch.cuda.synchronize() (the driver's form)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
dev = torch.device("cuda")
ln, d, B, K = bench.KAGGLE_LN, 36, 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 20
ev = bench.make_tables(ln, d)
bs = bench.make_batches(ln, B, 64, 1, dev, "uniform")
x = torch.rand(B, d, device=dev)
F = 27
out = torch.empty((B, d + F * (F - 1) // 2), device=dev)
step = lambda i: E.apply_emb_interact(x, bs[i % 64][0], bs[i % 64][1], ev, out=out, one_index_per_bag=True)
for i in range(50): step(i)
t_s = time.perf_counter()
while time.perf_counter() - t_s < 0.35:
    for i in range(50): step(i)
    torch.cuda.synchronize()
def timed(fn, reps=7):
    r = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / K * 1e6)
    print(' '.join('%.2f' % v for v in r))
    return sorted(r)
eager = timed(lambda: [step(i) for i in range(K)])
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3): step(i)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    for i in range(K): step(i)
g.replay(); torch.cuda.synchronize()
graph = timed(lambda: g.replay())
print("K=%d  eager us/step: min %.2f med %.2f | graph us/step: min %.2f med %.2f" % (K, eager[0], eager[len(eager) // 2], graph[0], graph[len(graph) // 2]))
