#!/usr/bin/env python3
"""Differential fuzz on the GPU: random shapes / precisions / bag structures through the fused kernel (offsets given,
offsets == NULL where legal), the two-kernel path and the oracle.  usage: python tools/fuzz.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import evstore_dlrm_amd as E  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)


def one_case(rs, case):
    d = int(rs.choice([16, 32, 36, 48, 64, 128]))
    T = int(rs.choice([1, 2, 6, 7, 8, 13, 14, 15, 16, 17, 20, 21, 26, 27, 28, 31]))
    B = int(rs.choice([1, 2, 3, 4, 5, 63, 64, 65, 257, 1000, 4099, 9000]))
    codec = int(rs.choice([32, 32, 16, 8, 4]))
    itself = bool(rs.randint(0, 2))
    mode = rs.choice(["arange", "ragged", "ragged", "empty-heavy", "moved"])
    weighted = codec == 32 and rs.randint(0, 4) == 0 and mode != "arange"
    ln = [int(rs.choice([1, 2, 3, 17, 300, 5000])) for _ in range(T)]
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in ln]
    raws = [orc.encode_table(t, codec) for t in tabs]
    ev = E.EVTables([torch.from_numpy(r).cuda() for r in raws], d, codec)
    if mode == "arange":
        lens = np.ones((T, B), dtype=np.int64)
    elif mode == "moved":      # nnz == B, a few units moved between bags: the in-kernel offsets check and its slow loop
        lens = np.ones((T, B), dtype=np.int64)
        for _ in range(int(rs.choice([1, 3, 40]))):
            k, src_b, dst_b = rs.randint(0, T), rs.randint(0, B), rs.randint(0, B)
            if lens[k, src_b] > 0:
                lens[k, src_b] -= 1
                lens[k, dst_b] += 1
    elif mode == "ragged":
        lens = rs.randint(0, 4, size=(T, B))
    else:
        lens = (rs.rand(T, B) < 0.15).astype(np.int64) * rs.randint(1, 6, size=(T, B))
    lS_i_np = [rs.randint(0, ln[k], size=int(lens[k].sum())).astype(np.int64) for k in range(T)]
    lS_o_np = [np.concatenate([[0], np.cumsum(lens[k])[:-1]]).astype(np.int64) for k in range(T)]
    w_np = [rs.uniform(0.5, 1.5, size=ln[k]).astype(np.float32) if weighted and rs.randint(0, 2) else None for k in range(T)]
    if not any(w is not None for w in w_np):
        w_np = None
    pad = int(rs.choice([0, 0, 4, 28]))      # x as a row-strided view of a wider buffer (the bottom MLP's output tile)
    xw = rs.uniform(-1, 1, size=(B, d + pad)).astype(np.float32)
    x_np = np.ascontiguousarray(xw[:, :d])
    x = torch.from_numpy(xw).cuda()[:, :d]
    lS_i = [torch.from_numpy(a).cuda() for a in lS_i_np]
    lS_o = [torch.from_numpy(a).cuda() for a in lS_o_np]
    w = None if w_np is None else [None if a is None else torch.from_numpy(a).cuda() for a in w_np]
    tag = "case %d: d=%d T=%d B=%d codec=%d itself=%s mode=%s weighted=%s" % (case, d, T, B, codec, itself, mode, w is not None)
    fused = E.apply_emb_interact(x, lS_o, lS_i, ev, w, itself, check_indices=True)
    ly = E.apply_emb(lS_o, lS_i, ev, w, lazy=False)
    two = E.interact_features(x, ly, "dot", itself)
    i8 = codec == 8 and d == 36 and T + 1 > 16   # (round 5: the fused u8 launch multiplies row x row on the integer matrix pipe: the tolerance, not the bits)
    if i8:
        torch.testing.assert_close(fused, two, rtol=1e-5, atol=2e-6 * max(1.0, float(two.abs().max())), msg=tag + ": fused vs two-kernel")
    else:
        assert torch.equal(fused, two), tag + ": fused != two-kernel"
    if mode == "arange" and w is None:
        st_i = torch.stack(lS_i)
        st_o = torch.stack(lS_o)
        a = E.apply_emb_interact(x, st_o, st_i, ev, None, itself, one_index_per_bag=True)
        b = E.apply_emb_interact(x, st_o, st_i, ev, None, itself)
        if i8:   # (small batches with lS_o given take the general loop -- fp32 chains --, the declared form the rows-in-registers kernel)
            torch.testing.assert_close(a, b, rtol=1e-5, atol=2e-6 * max(1.0, float(fused.abs().max())), msg=tag)
            torch.testing.assert_close(a, fused, rtol=1e-5, atol=2e-6 * max(1.0, float(fused.abs().max())), msg=tag)
        else:
            assert torch.equal(a, fused) and torch.equal(b, fused), tag + ": stacked / one-index path differs"
    want_ly = orc.apply_emb(lS_o_np, lS_i_np, tabs if codec == 32 else raws, w_np, codec, d)
    for k in range(T):
        assert np.array_equal(ly[k].cpu().numpy().view(np.uint32), want_ly[k].view(np.uint32)), tag + ": pooled rows of table %d" % k
    want = orc.interact_features(x_np, want_ly, itself=itself) if itself else orc.interact_features(x_np, want_ly)
    # fp32 MFMA chains against the oracle's double accumulation: the absolute slack scales with the length of the dot
    # products and with the magnitude of the pooled rows (bags of up to maxlen rows in [-1, 1])
    maxlen = max(1, int(lens.max()))
    np.testing.assert_allclose(fused.cpu().numpy(), want, rtol=1e-5, atol=2e-6 * (d / 36.0) * maxlen * maxlen, err_msg=tag)
    return tag


def interact_case(rs, case):
    """interact_features over dense features with arbitrary row strides (views of a wider buffer), dot +- itself, cat."""
    d = int(rs.choice([16, 32, 36, 48, 64, 128, 20, 7]))
    F = int(rs.choice([2, 3, 9, 16, 17, 27, 32]))
    B = int(rs.choice([1, 2, 77, 1000, 5000]))
    itself = bool(rs.randint(0, 2))
    pad = int(rs.choice([0, 4, 12]))
    buf = torch.from_numpy(rs.uniform(-1, 1, size=(B, F, d + pad)).astype(np.float32)).cuda()
    feats = [buf[:, f, :d] for f in range(F)]
    tag = "interact case %d: d=%d F=%d B=%d itself=%s pad=%d" % (case, d, F, B, itself, pad)
    x_np = feats[0].cpu().numpy()
    ly_np = [f.cpu().numpy() for f in feats[1:]]
    R = E.interact_features(feats[0], feats[1:], "dot", itself)
    want = orc.interact_features(x_np, ly_np, itself=itself)
    np.testing.assert_allclose(R.cpu().numpy(), want, rtol=1e-5, atol=2e-6 * max(1.0, d / 36.0), err_msg=tag)
    assert torch.equal(R[:, :d], feats[0]), tag + ": x passthrough"
    cat = E.interact_features(feats[0], feats[1:], "cat")
    assert torch.equal(cat, torch.cat(feats, dim=1)), tag + ": cat"
    return tag


def tile_case(rs, case):
    """apply_emb straight into the (B,F,d) interaction tile, then interact_features over the tile's views."""
    d = int(rs.choice([16, 36, 64]))
    T = int(rs.choice([1, 5, 15, 16, 26]))
    B = int(rs.choice([1, 3, 200, 3000]))
    ln = [int(rs.choice([2, 50, 4000])) for _ in range(T)]
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in ln]
    ev = E.EVTables.from_fp32([torch.from_numpy(t) for t in tabs])
    lens = rs.randint(0, 3, size=(T, B))
    li = [rs.randint(0, ln[k], size=int(lens[k].sum())).astype(np.int64) for k in range(T)]
    lo = [np.concatenate([[0], np.cumsum(lens[k])[:-1]]).astype(np.int64) for k in range(T)]
    tag = "tile case %d: d=%d T=%d B=%d" % (case, d, T, B)
    x_np = rs.uniform(-1, 1, size=(B, d)).astype(np.float32)
    tile = torch.zeros((B, T + 1, d), device="cuda")
    tile[:, 0, :] = torch.from_numpy(x_np).cuda()
    ly = E.apply_emb([torch.from_numpy(a).cuda() for a in lo], [torch.from_numpy(a).cuda() for a in li], ev, None, out=tile)
    want_ly = orc.apply_emb(lo, li, tabs)
    for k in range(T):
        assert np.array_equal(tile[:, k + 1, :].cpu().numpy().view(np.uint32), want_ly[k].view(np.uint32)), tag + ": tile slot %d" % (k + 1)
    R = E.interact_features(tile[:, 0, :], ly)
    np.testing.assert_allclose(R.cpu().numpy(), orc.interact_features(x_np, want_ly), rtol=1e-5, atol=2e-6 * (d / 36.0) * 4, err_msg=tag)
    return tag


def sharded_case(rs, case):
    """The sharded op on virtual ranks (one GPU, the all-to-all done by block copies) against the single-rank result."""
    from evstore_dlrm_amd import sharded
    T = int(rs.choice([2, 6, 13, 26]))
    world = int(rs.choice([1, 2, 3, 4]))
    d = int(rs.choice([16, 36, 64]))
    Bl = int(rs.choice([1, 5, 48, 300]))
    Bg = world * Bl
    ln = [int(rs.choice([3, 40, 900, 20000, 90000])) for _ in range(T)]
    policy = rs.choice(["count", "rows", "rows+replicate", "hbm"])
    thr = int(rs.choice([10, 1000, 50000]))
    budget = int(rs.choice([0, 500, 30000, 10 ** 9]))
    bag1 = bool(rs.randint(0, 2))
    tag = "sharded case %d: T=%d world=%d d=%d Bl=%d policy=%s thr=%d budget=%d bag1=%s" % (case, T, world, d, Bl, policy, thr, budget, bag1)
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in ln]
    lens = np.ones((T, Bg), dtype=np.int64) if bag1 else rs.randint(0, 4, size=(T, Bg))
    lS_i = [torch.from_numpy(rs.randint(0, ln[k], size=int(lens[k].sum())).astype(np.int64)).cuda() for k in range(T)]
    lS_o = [torch.from_numpy(np.concatenate([[0], np.cumsum(lens[k])[:-1]]).astype(np.int64)).cuda() for k in range(T)]
    x = torch.from_numpy(rs.uniform(-1, 1, size=(Bg, d)).astype(np.float32)).cuda()
    want = E.apply_emb_interact(x, lS_o, lS_i, E.EVTables.from_fp32([torch.from_numpy(t) for t in tabs]))
    owner = sharded.plan_placement(ln, world, policy, replicate_max_rows=thr, replicate_budget_rows=budget)
    ops = []
    for r in range(world):
        held = {t: torch.from_numpy(tabs[t]) for t in range(T) if owner[t] in (r, -1)}
        ops.append(sharded.ShardedEmbeddingInteract(ln, d, r, world, held, sharded.HipBackend(torch.device("cuda")), policy=policy,
                                                    replicate_max_rows=thr, replicate_budget_rows=budget, one_index_per_bag=bag1))
        assert ops[-1].owner == owner, tag + ": ranks disagree on the placement"
    sends = [op.pool(lS_o, lS_i)[0] if len(op.my_own) else None for op in ops]   # the exchange itself is done by hand below
    torch.cuda.synchronize()
    for r, op in enumerate(ops):
        _, _, out_splits = op._splits(Bg)
        blocks = [sends[p][r * Bl:(r + 1) * Bl].reshape(-1) for p in range(world) if len(ops[p].my_own)]
        recv = torch.cat(blocks) if blocks else x.new_empty((0,))
        R = op.finish((None, recv, Bg, Bl, out_splits), x[r * Bl:(r + 1) * Bl], lS_o, lS_i)
        assert torch.equal(R, want[r * Bl:(r + 1) * Bl]), tag + ": rank %d" % r
    return tag


def p2p_case(rs, case):
    """The sharded op with exchange_mode "p2p" on virtual ranks (one process, the ranks wired to each other's receive buffers):
    random shapes / placements / slot orders, two or three rounds -- every rank's slice against the single-rank result."""
    from evstore_dlrm_amd import sharded
    T = int(rs.choice([2, 6, 13, 26]))
    world = int(rs.choice([1, 2, 3, 4, 8]))
    d = int(rs.choice([16, 36, 64]))
    Bl = int(rs.choice([1, 5, 16, 48, 130]))
    Bg = world * Bl
    ln = [int(rs.choice([3, 40, 900, 20000])) for _ in range(T)]
    policy = str(rs.choice(["count", "rows", "rows+replicate", "rowsplit"]))
    thr = int(rs.choice([10, 1000]))
    bag1 = bool(rs.randint(0, 2))
    tag = "p2p case %d: T=%d world=%d d=%d Bl=%d policy=%s thr=%d bag1=%s" % (case, T, world, d, Bl, policy, thr, bag1)
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in ln]
    ev_all = E.EVTables.from_fp32([torch.from_numpy(t) for t in tabs])
    owner = sharded.plan_placement(ln, world, policy, replicate_max_rows=thr)
    shared, ops = {}, []
    for r in range(world):
        held = {}
        for t in range(T):
            if owner[t] in (r, -1):
                held[t] = torch.from_numpy(tabs[t])
            elif owner[t] == -2:
                lo, hi = sharded.row_range(ln[t], r, world)
                held[t] = torch.from_numpy(np.ascontiguousarray(tabs[t][lo:hi]))
        op = sharded.ShardedEmbeddingInteract(ln, d, r, world, held, sharded.HipBackend(torch.device("cuda")), policy=policy,
                                              replicate_max_rows=thr, one_index_per_bag=bag1)
        op.exchange_mode, op.p2p_virtual = "p2p", shared
        ops.append(op)
    try:
        for op in ops:
            if op.any_sharded:
                op._p2p_state(Bg)
        for rnd in range(int(rs.choice([2, 3]))):
            lens = np.ones((T, Bg), dtype=np.int64) if bag1 else rs.randint(0, 4, size=(T, Bg))
            lS_i = [torch.from_numpy(rs.randint(0, ln[k], size=int(lens[k].sum())).astype(np.int64)).cuda() for k in range(T)]
            lS_o = [torch.from_numpy(np.concatenate([[0], np.cumsum(lens[k])[:-1]]).astype(np.int64)).cuda() for k in range(T)]
            x = torch.from_numpy(rs.uniform(-1, 1, size=(Bg, d)).astype(np.float32)).cuda()
            want = E.apply_emb_interact(x, lS_o, lS_i, ev_all)
            hs = [op.start(lS_o, lS_i, slot=rnd % 2) for op in ops]
            for op in ops:
                op.p2p_flush()
            Rs = [op.finish(hs[r], x[r * Bl:(r + 1) * Bl], lS_o, lS_i) for r, op in enumerate(ops)]
            for op in ops:
                op.p2p_flush()
            assert E._lib.lib().evs_check_index_errors(None) == 0, tag + ": a hand-over did not arrive"
            for r in range(world):
                if policy == "rowsplit" and not bag1:   # (partials of a split bag are summed in rank order: not the single launch's order)
                    assert torch.allclose(Rs[r], want[r * Bl:(r + 1) * Bl], rtol=1e-5, atol=1e-5), tag + ": rank %d round %d" % (r, rnd)
                else:
                    assert torch.equal(Rs[r], want[r * Bl:(r + 1) * Bl]), tag + ": rank %d round %d" % (r, rnd)
    finally:
        torch.cuda.synchronize()
        for op in ops:
            for st in op._p2p.values():
                st.close()
    return tag


def encode_case(rs, case):
    d = int(rs.choice([2, 16, 36, 64]))
    n = int(rs.choice([1, 7, 500]))
    kind = rs.choice(["uniform", "normal", "wide", "tiny"])
    w = {"uniform": lambda: rs.uniform(-1, 1, size=(n, d)), "normal": lambda: rs.standard_normal(size=(n, d)) * 0.4,
         "wide": lambda: rs.uniform(-1.3, 1.3, size=(n, d)), "tiny": lambda: rs.standard_normal(size=(n, d)) * 1e-4}[kind]().astype(np.float32)
    ev = E.EVTables.from_fp32([torch.from_numpy(w)])
    tag = "encode case %d: d=%d n=%d %s" % (case, d, n, kind)
    for bits in (16, 8, 4):
        got = ev.encode(bits).raw[0].cpu().numpy()
        assert np.array_equal(got, orc.encode_table(w, bits)), tag + ": %d-bit codes" % bits
    return tag


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    t0 = time.time()
    n = 0
    last = ""
    kinds = (one_case, one_case, one_case, interact_case, tile_case, sharded_case, p2p_case, encode_case)
    while time.time() - t0 < seconds:
        fn = kinds[int(rs.randint(0, len(kinds)))]
        if os.environ.get("EVS_FUZZ_VERBOSE"):
            print("-> case %d %s" % (n, fn.__name__), flush=True)
        last = fn(rs, n)
        n += 1
    print("fuzz ok: %d cases in %.0f s (seed %d); last %s" % (n, time.time() - t0, seed, last))


if __name__ == "__main__":
    main()
