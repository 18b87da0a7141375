"""Developer probe: where a request of the exact batch-1 GPU engine spends its time -- a library built with -DEVS_X_EXACT_TIMING
(tools/variants.sh xt@evs_cache:"-DEVS_X_EXACT_TIMING"; EVS_LIB_PATH=.../libevstore_hip_xt.so) sums 100 MHz ticks per stage over a launch."""
import sys, os, time, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
dev = torch.device("cuda"); T, d = 26, 36
ev = bench.make_tables(bench.KAGGLE_LN, d)
b = bench.make_batches(bench.KAGGLE_LN, 2048, 3, seed=3, device=dev, dist="zipf", alpha=1.05)
L = E._lib.lib()
f = ctypes.CDLL(os.environ["EVS_LIB_PATH"]).evs_x_exact_ticks
c1 = E.GpuCache("evlfu", 200000, T, d, 32, "python", dev); c1.set_backing(ev)
rq = [x[1].t().contiguous().to(torch.int32) for x in b]
c1.request(rq[0]); torch.cuda.synchronize()
out = (ctypes.c_longlong * 8)()
f(out, 1)
t0 = time.perf_counter(); c1.request(rq[1]); c1.request(rq[2]); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4096 * 1e6
f(out, 0)
names = ["request read", "probe", "prefetch", "lane-0 loop", "rows out", "fills", "-", "-"]
print("exact replay: %.2f us per request; per stage (us): %s" % (dt, ", ".join("%s %.2f" % (names[k], out[k] / 100.0 / 4096) for k in range(6))))
print("inside the loop (us per request): hits visited %.2f, misses %.2f" % (out[6] / 100.0 / 4096, out[7] / 100.0 / 4096))
print(c1.stats())
