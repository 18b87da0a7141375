"""Developer micro-benchmark: the exact batch-1 GPU engine per synchronised request (copy vs pinned in/out buffers)."""
import sys, time, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
dev = torch.device("cuda")
T, d = 26, 36
ev = bench.make_tables(bench.KAGGLE_LN, d)
batches = bench.make_batches(bench.KAGGLE_LN, 64, 24, seed=3, device=dev, dist="zipf", alpha=0.75)
req1 = torch.cat([b[1].t().contiguous().to(torch.int32) for b in batches])[:1500].contiguous()
host_rows = req1.cpu()
for mode in ("copy", "pinned", "copy", "pinned"):
    c1 = E.GpuCache("evlfu", 200000, T, d, 32, "python", dev); c1.set_backing(ev)
    pin_rows = torch.empty((1, T), dtype=torch.int32).pin_memory()
    pin_out = torch.empty((1, T, d), dtype=torch.float32).pin_memory()
    pin_hit = torch.empty((1, T), dtype=torch.uint8).pin_memory()
    o1 = torch.empty((1, T, d), device=dev); h1 = torch.empty((1, T), dtype=torch.uint8, device=dev)
    lat = []
    for i in range(1500):
        t1 = time.perf_counter()
        if mode == "copy":
            rq = host_rows[i:i + 1].to(dev, non_blocking=True)
            c1.request(rq, out=o1, hit=h1)
            pin_out.copy_(o1, non_blocking=True)
        else:
            pin_rows.copy_(host_rows[i:i + 1])
            c1.request(pin_rows, out=pin_out, hit=pin_hit)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e6)
    print(mode, "p50 %.1f p95 %.1f" % (np.percentile(lat[200:], 50), np.percentile(lat[200:], 95)))
