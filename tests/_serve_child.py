"""Child process of tests/test_gpu_serve.py: the resident dispatcher of the fused launch (evs_emb_interact_serve_*, round 6) against the
launched kernel -- same bits for every batch size from 1 to more chunks than the grid has workers, batches posted back to back,
ragged bags (the block's slow loop), the grid leaving idle and coming back, stop(), out-of-range indices flagged."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import evstore_dlrm_amd as E  # noqa: E402


def case(ln, d, sizes, n_blocks=0, seed=0):
    dev = torch.device("cuda")
    rs = np.random.RandomState(seed)
    T = len(ln)
    ws = [torch.from_numpy(rs.uniform(-1, 1, size=(n, d)).astype(np.float32)).to(dev) for n in ln]
    ev = E.EVTables.from_fp32(ws)
    srv = E.InteractServer(ev, n_blocks=n_blocks, idle_us=150)
    for B in sizes:
        idx = torch.stack([torch.from_numpy(rs.randint(0, n, size=B)) for n in ln]).to(dev)
        off = torch.arange(B, device=dev).repeat(T, 1)
        x = torch.rand(B, d, device=dev)
        want = E.apply_emb_interact(x, off, idx, ev)
        torch.cuda.synchronize()
        got = srv(x, off, idx)
        torch.cuda.synchronize()
        assert torch.equal(got, want), ("d=%d T=%d B=%d" % (d, T, B), (got - want).abs().max().item())
    return srv, ev, ws, rs


def main():
    dev = torch.device("cuda")
    ln26 = [5000, 7, 2600, 40, 9000, 3, 12000] + [100] * 19
    # every size class: fewer samples than a chunk, ragged last chunk, one generation of blocks, more chunks than workers
    srv, ev, ws, rs = case(ln26, 36, [1, 5, 16, 17, 128, 2048, 2050, 16384, 40000])
    T, d = 26, 36
    # ---- 100 small batches posted back to back, then waited for in order; and in reverse order
    B = 256
    xs = [torch.rand(B, d, device=dev) for _ in range(100)]
    idxs = [torch.stack([torch.from_numpy(rs.randint(0, n, size=B)) for n in ln26]).to(dev) for _ in range(100)]
    off = torch.arange(B, device=dev).repeat(T, 1)
    wants = [E.apply_emb_interact(xs[i], off, idxs[i], ev) for i in range(100)]
    torch.cuda.synchronize()
    for order in (1, -1):
        outs = [torch.zeros_like(wants[0]) for _ in range(100)]
        tickets = [srv.post(xs[i], off, idxs[i], out=outs[i])[0] for i in range(100)]
        for i in list(range(100))[::order]:
            srv.wait(tickets[i])
        torch.cuda.synchronize()
        for i in range(100):
            assert torch.equal(outs[i], wants[i]), ("pipelined", order, i)
    # ---- inputs REWRITTEN IN PLACE between posts while the grid stays resident (no kernel boundary invalidates its caches: the
    # body reads x / indices / offsets through agent-scope loads): every batch must see the new contents
    xr, ir, orr = torch.empty(B, d, device=dev), torch.empty((T, B), dtype=torch.int64, device=dev), torch.empty((T, B), dtype=torch.int64, device=dev)
    outr = torch.empty_like(wants[0])
    for i in range(40):
        xr.copy_(xs[i]); ir.copy_(idxs[i]); orr.copy_(off)
        torch.cuda.synchronize()
        got = srv(xr, orr, ir, out=outr)
        assert torch.equal(got, wants[i]), ("inputs rewritten in place", i)
    # ---- the grid leaves idle (150 us) and the next post brings it back; stop() in between
    for pause in (0.002, 0.0, 0.01):
        time.sleep(pause)
        got = srv(xs[3], off, idxs[3])
        assert torch.equal(got, wants[3])
    srv.stop()
    got = srv(xs[4], off, idxs[4])
    assert torch.equal(got, wants[4])
    # ---- other launches between posts (the grid has to have left for them to run: stop(), or its idle time-out)
    srv.stop()
    y = torch.rand(1000, 1000, device=dev) @ torch.rand(1000, 1000, device=dev)
    torch.cuda.synchronize()
    got = srv(xs[5], off, idxs[5])
    assert torch.equal(got, wants[5]) and bool(torch.isfinite(y).all())
    # ---- ragged bags: offsets that are not arange -> the block's slow loop, general bag semantics (empty bags, several indices)
    B = 64
    lens = rs.randint(0, 3, size=(T, B))
    lens[:, -1] = 0
    # whole-batch form: nnz == B per table, so redistribute: every table's lengths sum to B
    for t in range(T):
        lens[t] = 1
        a, b = rs.randint(0, B, size=2)
        if a != b:
            lens[t, a] += 1; lens[t, b] -= 1
    offs = torch.from_numpy(np.stack([np.concatenate([[0], np.cumsum(lens[t])[:-1]]) for t in range(T)]).astype(np.int64)).to(dev)
    idx = torch.stack([torch.from_numpy(rs.randint(0, n, size=B)) for n in ln26]).to(dev)
    x = torch.rand(B, d, device=dev)
    want = E.apply_emb_interact(x, offs, idx, ev)
    torch.cuda.synchronize()
    got = srv(x, offs, idx)
    assert torch.equal(got, want), "ragged bags"
    # ---- an out-of-range index: skipped and flagged as the launch form does
    bad = idx.clone()
    bad[2, 7] = ln26[2] + 5
    arange = torch.arange(B, device=dev).repeat(T, 1)
    want = E.apply_emb_interact(x, arange, bad, ev)
    torch.cuda.synchronize()
    E._lib.lib().evs_check_index_errors(None)   # (clear what the launch form raised)
    got = srv(x, arange, bad)
    assert torch.equal(got, want)
    assert E._lib.lib().evs_check_index_errors(None) != 0, "the resident form must flag the bad index too"
    srv.close()
    # ---- other shapes: F <= 16 (one MFMA tile), d = 16 / 32 / 64, a small grid (chunks queue up behind few workers)
    for ln, dd, nb in (([300, 5, 1000, 64, 17, 900, 33, 2], 16, 0), ([300, 5, 1000] * 6, 32, 0), (ln26, 64, 0), (ln26, 36, 9)):
        s2 = case(ln, dd, [3, 200, 3000], n_blocks=nb, seed=dd + nb)[0]
        s2.close()
    print("SERVE_OK")


if __name__ == "__main__":
    main()
