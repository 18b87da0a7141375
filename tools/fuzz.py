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
    mode = rs.choice(["arange", "ragged", "ragged", "empty-heavy"])
    weighted = codec == 32 and rs.randint(0, 4) == 0 and mode != "arange"
    ln = [int(rs.choice([1, 2, 3, 17, 300, 5000])) for _ in range(T)]
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in ln]
    raws = [orc.encode_table(t, codec) for t in tabs]
    ev = E.EVTables([torch.from_numpy(r).cuda() for r in raws], d, codec)
    if mode == "arange":
        lens = np.ones((T, B), dtype=np.int64)
    elif mode == "ragged":
        lens = rs.randint(0, 4, size=(T, B))
    else:
        lens = (rs.rand(T, B) < 0.15).astype(np.int64) * rs.randint(1, 6, size=(T, B))
    lS_i_np = [rs.randint(0, ln[k], size=int(lens[k].sum())).astype(np.int64) for k in range(T)]
    lS_o_np = [np.concatenate([[0], np.cumsum(lens[k])[:-1]]).astype(np.int64) for k in range(T)]
    w_np = [rs.uniform(0.5, 1.5, size=ln[k]).astype(np.float32) if weighted and rs.randint(0, 2) else None for k in range(T)]
    if not any(w is not None for w in w_np):
        w_np = None
    x_np = rs.uniform(-1, 1, size=(B, d)).astype(np.float32)
    x = torch.from_numpy(x_np).cuda()
    lS_i = [torch.from_numpy(a).cuda() for a in lS_i_np]
    lS_o = [torch.from_numpy(a).cuda() for a in lS_o_np]
    w = None if w_np is None else [None if a is None else torch.from_numpy(a).cuda() for a in w_np]
    tag = "case %d: d=%d T=%d B=%d codec=%d itself=%s mode=%s weighted=%s" % (case, d, T, B, codec, itself, mode, w is not None)
    fused = E.apply_emb_interact(x, lS_o, lS_i, ev, w, itself, check_indices=True)
    ly = E.apply_emb(lS_o, lS_i, ev, w, lazy=False)
    two = E.interact_features(x, ly, "dot", itself)
    assert torch.equal(fused, two), tag + ": fused != two-kernel"
    if mode == "arange" and w is None:
        st_i = torch.stack(lS_i)
        st_o = torch.stack(lS_o)
        a = E.apply_emb_interact(x, st_o, st_i, ev, None, itself, one_index_per_bag=True)
        b = E.apply_emb_interact(x, st_o, st_i, ev, None, itself)
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


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    t0 = time.time()
    n = 0
    last = ""
    while time.time() - t0 < seconds:
        last = one_case(rs, n)
        n += 1
    print("fuzz ok: %d cases in %.0f s (seed %d); last %s" % (n, time.time() - t0, seed, last))


if __name__ == "__main__":
    main()
