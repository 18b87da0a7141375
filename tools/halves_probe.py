"""Does ONE call that splits a 65 536-batch into two half-launches on two streams (fork / join by events or by signal words
around every call) beat the single launch?  (tools/dephase_probe.py measured the free-running two-stream form: +10 %.)
python tools/halves_probe.py [bits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
import evstore_dlrm_amd as E
from evstore_dlrm_amd import _lib

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d, T, B = 36, 26, 65536
ev = bench.make_tables(bench.KAGGLE_LN, d, bits=bits, codes="encoded" if bits != 32 else "random")
batches = bench.make_batches(bench.KAGGLE_LN, B, 4, 1, "cuda", "uniform")
x = torch.rand(B, d, device="cuda")
out = torch.empty(B, d + 351, device="cuda")
side = torch.cuda.Stream()
L = _lib.lib()
sig = []
for _ in range(2):
    p = C.c_void_p(); _lib.check(L.evs_signal_alloc(C.byref(p))); sig.append(p.value)
cnt = [0]
h = B // 2


def one(i, mode):
    off, idx = batches[i % 4]
    main = torch.cuda.current_stream()
    if mode == "single":
        E.apply_emb_interact(x, off, idx, ev, None, out=out, one_index_per_bag=True)
        return
    if mode == "events":
        ev0 = torch.cuda.Event(); ev0.record(main); side.wait_event(ev0)
    else:
        cnt[0] += 1
        L.evs_stream_write_value(main.cuda_stream, sig[0], cnt[0] & 0xffffffff)
        L.evs_stream_wait_value(side.cuda_stream, sig[0], cnt[0] & 0xffffffff)
    with torch.cuda.stream(side):
        E.apply_emb_interact(x[h:], off[:, :h], idx[:, h:], ev, None, out=out[h:], one_index_per_bag=True)
    E.apply_emb_interact(x[:h], off[:, :h], idx[:, :h], ev, None, out=out[:h], one_index_per_bag=True)
    if mode == "events":
        ev1 = torch.cuda.Event(); ev1.record(side); main.wait_event(ev1)
    else:
        L.evs_stream_write_value(side.cuda_stream, sig[1], cnt[0] & 0xffffffff)
        L.evs_stream_wait_value(main.cuda_stream, sig[1], cnt[0] & 0xffffffff)


def run(mode, n=100):
    for i in range(10):
        one(i, mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        one(i, mode)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for mode in ("single", "events", "signals", "single", "events", "signals"):
    print("u%d B=%d %s: %.1f us per batch" % (bits, B, mode, run(mode)), flush=True)
