"""apply_emb alone (one index per bag, Kaggle tables): stream time per call by table precision and batch size.
python tools/gather_bench.py [bits ...] [B=n ...]      (A/B by EVS_LIB_PATH)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import evstore_dlrm_amd as E
from tools.kbench import timeit, settle

bits_l = [int(a) for a in sys.argv[1:] if not a.startswith("B=")] or [32, 8]
Bs = [int(a[2:]) for a in sys.argv[1:] if a.startswith("B=")] or [16384, 65536]
d, T = 36, 26
for bits in bits_l:
    ev = bench.make_tables(bench.KAGGLE_LN, d, bits=bits, codes="encoded" if bits != 32 else "random")
    for B in Bs:
        batches = bench.make_batches(bench.KAGGLE_LN, B, 8, 1, "cuda", "uniform")
        out = torch.empty(T, B, d, device="cuda")
        f = lambda i: E.apply_emb(batches[i % 8][0], batches[i % 8][1], ev, None, lazy=False, _into=out)
        g = lambda i: E.apply_emb(batches[i % 8][0], batches[i % 8][1], ev, None, lazy=False, one_index_per_bag=True, _into=out)
        settle(f)
        tf, tg = timeit(f, 200), timeit(g, 200)
        mb = T * B * (d * bits // 8 + 8 + 4 * d) / 1e6
        print("u%-2d B=%6d: lS_o given %6.1f us | declared %6.1f us = %.2f TB/s algorithmic (%.0f MB, %.0f of them stores)"
              % (bits, B, tf, tg, mb / tg, mb, T * B * 4 * d / 1e6), flush=True)
    del ev
