#!/usr/bin/env python3
"""Developer micro-benchmark: the 26-table gather (evs_embedding_bag_sum) with offsets = arange against its offsets-free
row-gather form (offsets == NULL), (T,B,d) output, B = 16 384."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from evstore_dlrm_amd import _lib  # noqa: E402
from tools.kbench import timeit  # noqa: E402

ln, d, B = bench.KAGGLE_LN, 36, 16384
T = len(ln)
ev = bench.make_tables(ln, d)
batches = bench.make_batches(ln, B, 8, 1, "cuda", "uniform")
out = torch.empty((T, B, d), device="cuda")
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
tp = (C.c_void_p * T)(*ev._tables_c)
nr = (C.c_int64 * T)(*ln)
nz = (C.c_int64 * T)(*[B] * T)
off = torch.arange(B, dtype=torch.int64, device="cuda")
op = (C.c_void_p * T)(*[off.data_ptr()] * T)
ips = [(C.c_void_p * T)(*[b[1][k].data_ptr() for k in range(T)]) for b in batches]
for name, o in (("offsets = arange", op), ("offsets == NULL  ", None)):
    us = timeit(lambda i: _lib.check(L.evs_embedding_bag_sum(T, B, d, 32, tp, nr, ips[i % 8], o, nz, None, out.data_ptr(), B * d, d, st)), 300)
    print("gather 26 tables, %s: %.1f us" % (name, us))
