"""Shared by the distributed tests: the oracle standing in for the HIP kernels behind ShardedEmbeddingInteract's
backend hook (test infrastructure), fixture access, and a spawn helper for gloo runs on 127.0.0.1."""
import os
import socket

import numpy as np
import torch

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIST_CASES = ["dist_w2", "dist_w4", "dist_w2_kaggle", "dist_w4_kaggle", "dist_w2_itself"]


class OracleBackend:
    """test-only stand-in for sharded.HipBackend (same methods, numpy/oracle arithmetic)."""

    def make_tables(self, weights, d):
        return [np.ascontiguousarray(w.numpy()) for w in weights]

    def bag_sum_into(self, ev, table_ids_local, lS_o_rows, lS_i_rows, send, n_own, d, planned=False, bag1=False,
                     layout=None, row_lo=None, row_total=None):
        for j, k in enumerate(table_ids_local):
            idx, off = lS_i_rows[j].numpy(), lS_o_rows[j].numpy()
            if row_lo is None:
                pooled = orc.embedding_bag_sum(ev[k], idx, off)
            else:
                # this rank's rows [lo, lo + n) of a row_total-row table: indices of other ranks' rows contribute nothing
                lo, n = row_lo[j], ev[k].shape[0]
                assert ((idx >= 0) & (idx < row_total[j])).all()
                keep = (idx >= lo) & (idx < lo + n)
                B = off.shape[0]
                ends = np.concatenate([off[1:], [idx.shape[0]]])
                new_off = np.zeros(B, np.int64)
                cnt = np.array([keep[off[b]:ends[b]].sum() for b in range(B)], np.int64)
                new_off[1:] = np.cumsum(cnt)[:-1]
                tab = ev[k] if n else np.zeros((1, d), np.float32)
                pooled = orc.embedding_bag_sum(tab, idx[keep] - lo, new_off)
            if layout is None:
                send[:, j, :] = torch.from_numpy(pooled)
                continue
            B, foff, tstride, bstride, pstride, bpp = layout
            flat = send.view(-1)
            for b in range(B):
                o = foff + j * tstride + (b // bpp) * pstride + (b % bpp) * bstride if bpp and bpp < B else foff + j * tstride + b * bstride
                flat[o:o + d] = torch.from_numpy(pooled[b])

    def interact_mixed(self, x, specs, ev, d, itself, out=None, planned=False):
        B = x.shape[0]
        ly = []
        for s in specs:
            if s[0] == "dense":
                ly.append(s[1].numpy().copy())
            elif s[0] == "gathered":   # a row-split table: bags over the fp32 rows of a flat buffer, summed in index order
                _, rows, n_rows, idx, off, nnz, off_len = s
                tab = rows.numpy()[:n_rows * d].reshape(n_rows, d)
                idx = idx.numpy()
                assert ((idx >= 0) & (idx < n_rows)).all()
                if off is None:
                    ly.append(tab[idx[:B]].copy())
                else:
                    ly.append(orc.embedding_bag_sum(tab, idx[:nnz], off.numpy()[:B]))
            else:
                _, k, idx, off, nnz, off_len = s
                if off is None:  # one index per bag: bag b = idx[b]
                    ly.append(ev[k][idx.numpy()[:B]].copy())
                    continue
                off = off.numpy()
                assert off_len == off.shape[0]
                # slice semantics of evs_feature.offsets_len: bag b ends at off[b+1] while it exists
                ends = np.concatenate([off[1:], [nnz]])[:B]
                starts = off[:B]
                rows = np.zeros((B, d), np.float32)
                for b in range(B):
                    rows[b] = orc.embedding_bag_sum(ev[k], idx.numpy()[starts[b]:ends[b]], np.array([0]))[0] \
                        if ends[b] > starts[b] else 0
                ly.append(rows)
        return torch.from_numpy(orc.interact_features(x.numpy(), ly, itself))


def load_dist(name):
    """-> dict with world, ln_emb, d, Bg, itself, tables, lS_o (T,Bg), lS_i (list), X, mlp (list), per-rank records."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    ln = [int(n) for n in g["ln_emb"]]
    d = int(g["m_spa"])
    tabs, p = [], 0
    for n in ln:
        tabs.append(np.ascontiguousarray(g["tables_cat"][p:p + n * d].reshape(n, d)))
        p += n * d
    lS_i, p = [], 0
    for n in g["lS_i_nnz"]:
        lS_i.append(np.ascontiguousarray(g["lS_i_cat"][p:p + int(n)]))
        p += int(n)
    world = int(g["world"])
    ranks = [{k: g["r%d_%s" % (r, k)] for k in ("ly_before", "block_cols", "blocks_after", "x", "R", "Z", "local_emb")}
             for r in range(world)]
    mlp = [g["mlp_%d" % i] for i in range(len([f for f in g.files if f.startswith("mlp_")]))]
    return {"world": world, "ln_emb": ln, "d": d, "Bg": int(g["B"]), "itself": bool(g["itself"]), "tables": tabs,
            "lS_o": g["lS_o"], "lS_i": lS_i, "X": g["X"], "mlp": mlp, "ranks": ranks,
            "n_emb_per_rank": [int(v) for v in g["n_emb_per_rank"]]}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn(world, target, *args, timeout=240):
    """Run target(rank, world, port, *args, q) in `world` spawned processes; returns the queue items sorted by rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    return sorted(res, key=lambda t: t[0])


def init_gloo(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
