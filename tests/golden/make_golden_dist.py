#!/usr/bin/env python3
"""Golden vectors of the reference's DISTRIBUTED forward (last row).

Runs only in the build container (needs /root/reference).  Spawns `world` processes, each IMPORTS the reference,
initialises its extend_distributed under gloo (extend_distributed.py:65-191) and runs
DLRM_Net.distributed_forward (dlrm_s_pytorch.py:529-586) -- both as the reference's own method (final Z) and
step by step with the reference's own functions, so that the per-rank intermediates are recorded:

  ly_before[rank]   what apply_emb returned on that rank: its LOCAL tables x the FULL batch
  blocks_after[rank] what ext_dist.alltoall(ly, n_emb_per_rank).wait() returned: one (B_local, T_p*d) block per
                    source rank p, stored concatenated along dim 1 = (B_local, T*d) in table order
  x[rank], R[rank], Z[rank]   bottom-MLP output, interact_features output, top-MLP output of the rank's batch slice

Every rank holds the same full tables / MLP weights in the fixture (a rank's model gets the rows of its slice
assigned after construction), so single-process code can check itself against the same file.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dist.py
"""
import os
import socket
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

CASES = [
    # name, world, ln_emb, d, global batch, max indices per bag, seed, itself
    ("dist_w2", 2, [50, 3, 4000, 17, 2500, 9, 1200], 16, 12, 4, 41, False),      # uneven tables 4/3
    ("dist_w4", 4, [50, 3, 4000, 17, 2500, 9, 1200], 16, 16, 3, 43, False),      # 2/2/2/1
    ("dist_w2_kaggle", 2, "kaggle200", 36, 16, 1, 45, False),                      # 26 tables, one index per bag, 13/13
    ("dist_w4_kaggle", 4, "kaggle200", 36, 32, 1, 47, False),                      # 7/7/6/6
    ("dist_w2_itself", 2, [64, 5, 777, 31, 12], 32, 10, 5, 49, True),             # diagonal kept, 3/2
    # (world 3 is not a reference configuration for this path: its all_to_all_single probe sends 4 elements,
    #  extend_distributed.py:163-168, which 3 ranks cannot split, so it falls back to scatter lists)
]
KAGGLE_LN = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593,
             3194, 27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def _inputs(ln_emb, d, Bg, n_idx, seed):
    """Full tables, MLP weights and one full batch (the same on every rank)."""
    rs = np.random.RandomState(seed)
    T = len(ln_emb)
    tabs = [rs.uniform(-np.sqrt(1.0 / n), np.sqrt(1.0 / n), size=(n, d)).astype(np.float32) for n in ln_emb]
    X = rs.rand(Bg, 13).astype(np.float32)
    if n_idx == 1:
        lens = np.ones((T, Bg), np.int64)
    else:
        lens = rs.randint(0, n_idx + 1, size=(T, Bg))
    lS_i = [rs.randint(0, ln_emb[k], size=int(lens[k].sum())).astype(np.int64) for k in range(T)]
    lS_o = np.stack([np.concatenate([[0], np.cumsum(lens[k])[:-1]]) for k in range(T)]).astype(np.int64)
    return rs, tabs, X, lS_o, lS_i


def _worker(rank, world, port, case, q):
    name, _, ln_emb, d, Bg, n_idx, seed, itself = case
    if ln_emb == "kaggle200":
        ln_emb = [min(n, 200) for n in KAGGLE_LN]
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank),
                       "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank)})
    import io
    import contextlib
    import torch
    sys.path.insert(0, HERE)
    import make_golden as G
    with contextlib.redirect_stdout(io.StringIO()):
        D = G.import_reference()[0]
    ext = sys.modules["extend_distributed"]
    import builtins
    builtins.print = ext.orig_print          # the reference replaces print by a rank-0-only one
    with contextlib.redirect_stdout(io.StringIO()):
        ext.init_distributed(rank=rank, size=world, backend="gloo")
    assert ext.my_size == world and ext.my_rank == rank and ext.alltoall_supported
    T = len(ln_emb)
    F = T + 1
    n_int = F * (F + 1) // 2 if itself else F * (F - 1) // 2
    ln_bot = np.array([13, 32, d])
    ln_top = np.array([n_int + d, 16, 1])
    rs, tabs, X, lS_o, lS_i = _inputs(ln_emb, d, Bg, n_idx, seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    dlrm = D.DLRM_Net(d, np.asarray(ln_emb), ln_bot, ln_top, arch_interaction_op="dot",
                      arch_interaction_itself=itself, sigmoid_bot=-1, sigmoid_top=ln_top.size - 2, ndevices=-1)
    # same weights on every rank (create_emb drew only this rank's tables, so its RNG streams differ per rank)
    mlp = []
    for seq, ln in ((dlrm.bot_l, ln_bot), (dlrm.top_l, ln_top)):
        for layer in seq:
            if hasattr(layer, "weight"):
                W = rs.normal(0, 0.3, size=tuple(layer.weight.shape)).astype(np.float32)
                b = rs.normal(0, 0.1, size=tuple(layer.bias.shape)).astype(np.float32)
                layer.weight.data = torch.tensor(W)
                layer.bias.data = torch.tensor(b)
                mlp += [W, b]
    assert len(dlrm.emb_l) == len(dlrm.local_emb_indices)
    for j, t in enumerate(dlrm.local_emb_indices):
        dlrm.emb_l[j].weight.data = torch.tensor(tabs[t])
    Xt = torch.tensor(X)
    lS_o_t = torch.tensor(lS_o)
    lS_i_t = [torch.tensor(v) for v in lS_i]
    with torch.no_grad():
        # the reference's own method, end to end
        Z_ref = dlrm(Xt, lS_o_t, lS_i_t)
        # and its body step by step (dlrm_s_pytorch.py:543-577) with the reference's own functions
        xs = Xt[ext.get_my_slice(Bg)]
        lo = lS_o_t[dlrm.local_emb_slice]
        li = lS_i_t[dlrm.local_emb_slice]
        ly = dlrm.apply_emb(lo, li, dlrm.emb_l, dlrm.v_W_l)
        req = ext.alltoall(ly, dlrm.n_emb_per_rank)
        x = dlrm.apply_mlp(xs, dlrm.bot_l)
        blocks = list(req.wait())
        R = dlrm.interact_features(x, blocks)
        Z = dlrm.apply_mlp(R, dlrm.top_l)
    assert torch.equal(Z, Z_ref)
    q.put((rank, {
        "ly_before": np.stack([v.numpy() for v in ly]) if len(ly) else np.zeros((0, Bg, d), np.float32),
        "block_cols": np.asarray([b.shape[1] for b in blocks], np.int64),
        "blocks_after": np.concatenate([b.numpy() for b in blocks], axis=1),
        "x": x.numpy(), "R": R.numpy(), "Z": Z.numpy(),
        "local_emb": np.asarray(dlrm.local_emb_indices, np.int64),
        "n_emb_per_rank": np.asarray(dlrm.n_emb_per_rank if dlrm.n_emb_per_rank else [T // world] * world, np.int64),
    }, mlp if rank == 0 else None))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def gen_case(case):
    import torch.multiprocessing as mp
    name, world, ln_emb, d, Bg, n_idx, seed, itself = case
    ln = [min(n, 200) for n in KAGGLE_LN] if ln_emb == "kaggle200" else ln_emb
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    _, tabs, X, lS_o, lS_i = _inputs(ln, d, Bg, n_idx, seed)
    out = {"world": np.int64(world), "ln_emb": np.asarray(ln, np.int64), "m_spa": np.int64(d), "B": np.int64(Bg),
           "itself": np.int64(itself), "seed": np.int64(seed), "X": X, "lS_o": lS_o,
           "lS_i_cat": np.concatenate(lS_i), "lS_i_nnz": np.asarray([v.size for v in lS_i], np.int64),
           "tables_cat": np.concatenate([t.reshape(-1) for t in tabs]),
           "n_emb_per_rank": res[0][1]["n_emb_per_rank"]}
    for i, w in enumerate(res[0][2]):
        out["mlp_%d" % i] = w
    for r, rec, _ in res:
        for k in ("ly_before", "block_cols", "blocks_after", "x", "R", "Z", "local_emb"):
            out["r%d_%s" % (r, k)] = rec[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "world", world, "tables/rank", out["n_emb_per_rank"].tolist(),
          "R", res[0][1]["R"].shape)


def main():
    only = sys.argv[1:]
    for c in CASES:
        if not only or c[0] in only:
            gen_case(c)


if __name__ == "__main__":
    main()
