"""CPU / gloo, world_size 2 and 3: the table-sharded path (placement, all-to-all splits, receive-block
feature pointers, batch-slice lookups of replicated tables) against the single-process oracle.
The oracle stands in for the HIP kernels through the backend hook of ShardedEmbeddingInteract."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from _dist_helpers import OracleBackend

LN = [50, 3, 4000, 17, 2500, 9, 1200]
D = 16


def _data(seed, Bg):
    rs = np.random.RandomState(seed)
    tabs = [rs.uniform(-1, 1, size=(n, D)).astype(np.float32) for n in LN]
    lens = rs.randint(0, 4, size=(len(LN), Bg))
    lS_i = [rs.randint(0, LN[k], size=lens[k].sum()).astype(np.int64) for k in range(len(LN))]
    lS_o = [np.concatenate([[0], np.cumsum(lens[k])[:-1]]).astype(np.int64) for k in range(len(LN))]
    x = rs.uniform(-1, 1, size=(Bg, D)).astype(np.float32)
    return tabs, lS_o, lS_i, x


def _worker(rank, world, port, policy, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import evstore_dlrm_amd as E
        from evstore_dlrm_amd import sharded
        Bg = 6 * world
        tabs, lS_o, lS_i, x = _data(3, Bg)
        budget = {"hbm": sum(LN), "hbm-partial": 150}.get(policy)   # everything replicated / only the small tables
        pol = "hbm" if policy.startswith("hbm") else policy
        owner = sharded.plan_placement(LN, world, pol, replicate_max_rows=100, replicate_budget_rows=budget)
        held = {t: torch.from_numpy(tabs[t]) for t in range(len(LN)) if owner[t] in (rank, -1)}
        op = sharded.ShardedEmbeddingInteract(LN, D, rank, world, held, OracleBackend(), policy=pol,
                                              replicate_max_rows=100, replicate_budget_rows=budget)
        Bl = Bg // world
        R = op.forward(torch.from_numpy(x[rank * Bl:(rank + 1) * Bl]), [torch.from_numpy(o) for o in lS_o],
                       [torch.from_numpy(i) for i in lS_i])
        ly = orc.apply_emb(lS_o, lS_i, tabs)
        want = orc.interact_features(x, ly)[rank * Bl:(rank + 1) * Bl]
        ok = np.allclose(R.numpy(), want, rtol=1e-6, atol=1e-6)
        q.put((rank, bool(ok), owner))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,policy", [(2, "count"), (2, "rows"), (2, "rows+replicate"), (3, "rows+replicate"),
                                          (2, "hbm"), (2, "hbm-partial")])
def test_sharded_forward_matches_single_process(world, policy):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, policy, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    owner = res[0][2]
    if policy == "rows+replicate":
        assert all((o == -1) == (LN[t] <= 100) for t, o in enumerate(owner))
    if policy == "hbm":          # the whole model fits the budget: no table is sharded, no exchange happens
        assert all(o == -1 for o in owner)
    if policy == "hbm-partial":  # smallest tables first while they fit 150 rows
        rep = sorted(LN[t] for t, o in enumerate(owner) if o == -1)
        assert rep and sum(rep) <= 150 and any(o >= 0 for o in owner)


def test_placement_policies():
    from evstore_dlrm_amd import sharded
    from bench import KAGGLE_LN
    # the reference's contiguous split: 26 tables over 8 ranks -> 4,4,3,3,3,3,3,3
    own = sharded.plan_placement(KAGGLE_LN, 8, "count")
    assert [own.count(r) for r in range(8)] == [4, 4, 3, 3, 3, 3, 3, 3] and own == sorted(own)
    rows = sharded.plan_placement(KAGGLE_LN, 8, "rows")
    load = [sum(n for n, o in zip(KAGGLE_LN, rows) if o == r) for r in range(8)]
    assert max(load) == 10131227  # one giant table alone on a rank: row balance != work balance
    rep = sharded.plan_placement(KAGGLE_LN, 8, "rows+replicate")
    assert sum(1 for o in rep if o >= 0) == 5 and sum(KAGGLE_LN[t] for t, o in enumerate(rep) if o == -1) < 600000
    # memory-aware: 64 GB per GPU holds the whole 4.9 GB model; 1 GB holds everything but the four largest tables
    assert all(o == -1 for o in sharded.plan_placement(KAGGLE_LN, 8, "hbm", replicate_budget_rows=int(64e9 / 144)))
    one_gb = sharded.plan_placement(KAGGLE_LN, 8, "hbm", replicate_budget_rows=int(1e9 / 144))
    assert sum(1 for o in one_gb if o >= 0) == 4 and sorted(o for o in one_gb if o >= 0) == [0, 1, 2, 3]


def test_ext_dist_helpers_match_reference_semantics():
    from evstore_dlrm_amd import extend_distributed as ext
    assert ext.get_my_slice(26, 0, 8) == slice(0, 4, 1) and ext.get_my_slice(26, 7, 8) == slice(23, 26, 1)
    assert ext.get_split_lengths(26, 1, 8) == (4, [4, 4, 3, 3, 3, 3, 3, 3])
    assert ext.get_split_lengths(16, 3, 8) == (2, None)
