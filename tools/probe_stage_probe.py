#!/usr/bin/env python3
"""Developer probe: where a block of the fused launch spends its time -- the plain launch (lS_o given) beside the cache tier's
one launch (probe + claim + interaction): per block, time from its entry to each stage, averaged over the blocks of many
launches.  Needs a library built with -DEVS_X_PT (tools/variants.sh pt@evs_fused_rf:"-DEVS_X_PT"; EVS_LIB_PATH=...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, evstore_dlrm_amd as E
f = ctypes.CDLL(os.environ["EVS_LIB_PATH"]).evs_x_pt
buf = (ctypes.c_ulonglong * 16)()
dev = torch.device("cuda"); ln, d, T, B = bench.KAGGLE_LN, 36, 26, 16384
ev = bench.make_tables(ln, d)
names = {1: "request rows in", 2: "set lines in", 3: "first barrier (agg)", 10: "claims sent", 11: "raises + claims looked at, tile written", 12: "second barrier", 5: "head over", 6: "row requests out", 8: "last stores out", 9: "stores acknowledged"}

def show(what):
    f(buf, 1)
    n = max(int(buf[15]), 1)
    print("%s: %d blocks; us from a block's entry: %s" % (what, n, ", ".join("%s %.2f" % (names[k], buf[k] / n / 100.0) for k in (1, 2, 3, 10, 11, 12, 5, 6, 8, 9) if buf[k])))

bs = bench.make_batches(ln, B, 8, seed=3, device=dev)
x = torch.rand((B, d), device=dev)
for i in range(20):
    E.apply_emb_interact(x, bs[i % 8][0], bs[i % 8][1], ev)
torch.cuda.synchronize(); f(buf, 1)
for i in range(100):
    E.apply_emb_interact(x, bs[i % 8][0], bs[i % 8][1], ev)
torch.cuda.synchronize(); show("plain launch (lS_o given)")
r = bench.cache_tier_section(ev, ln, d, B, dev, steps=100, batch1=False, settle_s=0.0)
show("cache tier, one launch (fill + %d timed batches; %.1f us per batch with the ticks in)" % (100, r["ms_per_step"] * 1e3))
