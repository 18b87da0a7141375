#!/usr/bin/env python3
"""Replay a recorded inference workload through the cache tier -- the reference's manual check
cache_algo/EvLFU_C1_Cython/test.py:10-59 (read the 26 workload-group-N.csv traces, EvLFU.cinit(768), cload_ev_tables, one crequest
per request, print the perfect-hit count and the wall time) on this package's modules.

    python tools/replay_workload.py <trace_dir> <ev_table_root> [--cache-size 768] [--algo evlfu_cython|evlfu|lru|lfu] [--batched B]

<trace_dir>: the directory write_inf_workload_to_file wrote; <ev_table_root>: where binary/ev-table-N.bin live (the storage
manager's MMAPFILEPY back-end).  --batched B replays B requests per call through the GPU tier's batched lookup (snapshot
semantics: its perfect-hit count is the batched policy's, not the sequential one's)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("ev_root")
    ap.add_argument("--cache-size", type=int, default=768)
    ap.add_argument("--algo", default="evlfu_cython", choices=["evlfu_cython", "evlfu", "lru", "lfu"])
    ap.add_argument("--ev-precs", type=int, default=32)
    ap.add_argument("--batched", type=int, default=0)
    a = ap.parse_args(argv)
    import evstore_dlrm_amd as E  # noqa: F401
    from evstore_dlrm_amd import evstore_utils
    from evstore_dlrm_amd.emb_storage import storage_manager as sm
    rows = evstore_utils.read_inf_workload(a.trace_dir)
    print(rows.shape)
    print("Done merging ALL workloads: total = ", rows.shape[0], 'rows')
    if a.batched > 0:
        import torch
        from evstore_dlrm_amd.dlrm_ops import EVTables
        ev = EVTables.from_bin_dir(os.path.join(a.ev_root, "binary"), codec=a.ev_precs)
        c = E.GpuCache({"evlfu_cython": "evlfu"}.get(a.algo, a.algo), a.cache_size, len(ev), ev.d, a.ev_precs, "cython" if a.algo == "evlfu_cython" else "python")
        c.set_backing(ev)
        perfect = 0
        torch.cuda.synchronize()
        t0 = time.time()
        for s in range(0, len(rows), a.batched):
            hit, _ = c.lookup_batch(torch.from_numpy(rows[s:s + a.batched]).cuda().contiguous())
            perfect += int(hit.all(1).sum())
        torch.cuda.synchronize()
        print("perfect hit:", perfect)
        print(time.time() - t0)
        return perfect
    sm.storage_type, sm.ev_precs = sm.EmbStorage.MMAPFILEPY, a.ev_precs
    sm.load_ev_table_into_emb_stor(a.ev_root)
    from evstore_dlrm_amd.cache_algo import EvLFU, EvLFU_C1, LFU, LRU
    if a.algo == "evlfu_cython":
        EvLFU.cinit(a.cache_size)
        EvLFU.cload_ev_tables()
        request = lambda ids: EvLFU.crequest(ids, False)[0]
    else:
        mod = {"evlfu": EvLFU_C1, "lru": LRU, "lfu": LFU}[a.algo]
        mod.init(a.cache_size)
        fn = {"evlfu": "request_to_ev_lfu", "lru": "request_to_lru", "lfu": "request_to_lfu"}[a.algo]
        request = lambda ids: getattr(mod, fn)(ids, False)[0]
    perfect = 0
    start_time = time.time()
    for group_row_ids in rows.tolist():
        if all(request(group_row_ids)):
            perfect += 1
    print("perfect hit:", perfect)
    print(time.time() - start_time)
    if a.algo == "evlfu_cython":
        EvLFU.cclose_ev_tables()
    sm.close_any_db_conn()
    return perfect


if __name__ == "__main__":
    main()
