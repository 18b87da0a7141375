"""Developer: does alternating independent batches over two streams hide the fused kernel's fill / drain?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
ev = bench.make_tables(bench.KAGGLE_LN, 36)
for B in (2048, 16384, 65536):
    bs = bench.make_batches(bench.KAGGLE_LN, B, 8, 1, "cuda", "uniform")
    x = torch.rand(B, 36, device="cuda")
    Rs = [torch.empty(B, 36 + 351, device="cuda") for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    for ns in (1, 2, 3):
        def run(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % ns]):
                    E.apply_emb_interact(x, bs[i % 8][0], bs[i % 8][1], ev, out=Rs[i % ns], one_index_per_bag=True)
        run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("B=%d streams=%d  %.1f us/step  %.2f G lookups/s" % (B, ns, dt / 400 * 1e6, 26 * B * 400 / dt / 1e9))
