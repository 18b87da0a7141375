#!/usr/bin/env python3
"""Developer probe: where the wall-clock of a 20-step timed region goes (host stamps around each bracket element)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import evstore_dlrm_amd as E  # noqa: E402

if os.environ.get("EVS_SPIN"):   # (developer A/B: how the host waits in a synchronise -- before anything creates the context)
    import ctypes
    print("hipSetDeviceFlags ->", ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(int(os.environ["EVS_SPIN"])))
torch.cuda.set_device(0)
B, d = 16384, 36
ev = bench.make_tables(bench.KAGGLE_LN, d, seed=0, device="cuda")
batches = bench.make_batches(bench.KAGGLE_LN, B, 64, seed=1, device="cuda")
x = torch.rand(B, d, device="cuda")
R = torch.empty(B, d + 351, device="cuda")


def step(i):
    o, ix = batches[i % 64]
    E.apply_emb_interact(x, o, ix, ev, None, out=R)


for i in range(3000):
    step(i)
torch.cuda.synchronize()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MODE = sys.argv[2] if len(sys.argv) > 2 else "event"
for rep in range(6):
    for i in range(300):
        step(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    t1 = time.perf_counter()
    step(0)
    t2 = time.perf_counter()
    for i in range(1, K):
        step(i)
    t3 = time.perf_counter()
    e1.record()
    t4 = time.perf_counter()
    if MODE == "event":
        while not e1.query():
            pass
    else:   # poll the STREAM: hipStreamQuery retires completed commands as it goes
        st = torch.cuda.current_stream()
        while not st.query():
            pass
    t5 = time.perf_counter()
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    ev_us = e0.elapsed_time(e1) * 1e3
    print("K=%d: e0.record %.1f | first launch %.1f | other launches %.1f | e1.record %.1f | poll %.1f | sync %.1f | wall %.1f us, events %.1f us (%.2f / step), wall - events %.1f"
          % (K, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t5 - t4) * 1e6, (t6 - t5) * 1e6, (t6 - t0) * 1e6, ev_us, ev_us / K, (t6 - t0) * 1e6 - ev_us))
