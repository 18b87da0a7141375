"""Developer micro-benchmark: the exact batch-1 GPU engine replaying 2 048 requests per launch (us per request)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
dev = torch.device("cuda"); T, d = 26, 36
ev = bench.make_tables(bench.KAGGLE_LN, d)
b = bench.make_batches(bench.KAGGLE_LN, 2048, 3, seed=3, device=dev, dist="zipf", alpha=1.05)
for rep in range(2):
    c1 = E.GpuCache("evlfu", 200000, T, d, 32, "python", dev); c1.set_backing(ev)
    rq = [x[1].t().contiguous().to(torch.int32) for x in b]
    c1.request(rq[0]); torch.cuda.synchronize()
    t0 = time.perf_counter(); c1.request(rq[1]); c1.request(rq[2]); torch.cuda.synchronize()
    print("exact replay: %.2f us per request" % ((time.perf_counter() - t0) / 4096 * 1e6))
