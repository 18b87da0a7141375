"""Child process of tests/test_p2p_exchange.py::test_p2p_exchange_between_processes: rank `r` of a `world`-process group whose
ranks ALL sit on GPU 0 (a test box has one GPU; RCCL refuses two ranks on one device -- the p2p exchange needs no RCCL: its
handles cross a gloo group, its data crosses IPC mappings).  Every rank runs the pipelined sharded steps with
exchange_mode = "p2p" and checks its R slice against the single-process fused launch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import evstore_dlrm_amd as E  # noqa: E402
from evstore_dlrm_amd import sharded  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if len(sys.argv) > 4 and sys.argv[4] == "bench":   # the N > 1 bench's side line (bench.py: exchange_p2p), every rank returns
        from types import SimpleNamespace
        args = SimpleNamespace(dim=36, batch=512, steps=20, warmup=3, placement="rows", replicate_gb=64.0, exchange_mode="inline")
        res = sharded.bench_p2p_side(args, [5000, 7, 2600, 40, 9000, 3, 12000] + [100] * 19, rank, world, dev)
        dist.barrier()
        dist.destroy_process_group()
        print("P2P_BENCH %s" % ("error: " + res["error"] if "error" in res else ("ok %.3f ms" % res["ms_per_step"] if "ms_per_step" in res else "other: %r" % (res,))))
        return
    rs = np.random.RandomState(3)                      # (the same model and batches in every process)
    ln = [5000, 7, 2600, 40, 9000, 3, 12000] + [100] * 19
    T, d, Bl = len(ln), 36, 96
    Bg = Bl * world
    ws = [torch.from_numpy(rs.uniform(-1, 1, size=(n, d)).astype(np.float32)).to(dev) for n in ln]
    ev = E.EVTables.from_fp32(ws)
    steps = 7
    idxs = [torch.stack([torch.from_numpy(rs.randint(0, n, size=Bg)) for n in ln]).to(dev) for _ in range(steps)]
    for k in range(steps):                             # the last and the first row of every table
        idxs[k][:, 0] = torch.tensor([n - 1 for n in ln], device=dev)
        idxs[k][:, -1] = 0
    off = torch.arange(Bg, device=dev).repeat(T, 1)
    x = torch.rand(Bg, d, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    wants = [E.apply_emb_interact(x, off, idxs[k], ev, one_index_per_bag=True) for k in range(steps)]
    sl = slice(rank * Bl, (rank + 1) * Bl)
    for policy in ("rows+replicate", "rows", "count", "rowsplit"):
        owner = sharded.plan_placement(ln, world, policy, replicate_max_rows=2000)
        held = {}
        for t in range(T):
            if owner[t] in (rank, -1):
                held[t] = ws[t]
            elif owner[t] == -2:
                lo, hi = sharded.row_range(ln[t], rank, world)
                held[t] = ws[t][lo:hi]
        op = sharded.ShardedEmbeddingInteract(ln, d, rank, world, held, sharded.HipBackend(dev), policy=policy, one_index_per_bag=True,
                                              replicate_max_rows=2000)
        op.exchange_mode = "p2p"
        lo_ = [off[t] for t in range(T)]
        outs = [torch.empty_like(wants[0][sl]) for _ in range(steps)]
        plans = [op.plan(x[sl], lo_, [idxs[k][t] for t in range(T)], out=outs[k], slot=k % 2) for k in range(steps)]
        # the bench's two-deep pipeline: the pool of step k + 1 is queued before the interaction of step k
        h = op.run_start(plans[0])
        for k in range(steps):
            nxt = op.run_start(plans[k + 1]) if k + 1 < steps else None
            op.run_finish(plans[k], h)
            h = nxt
        op.p2p_flush()
        torch.cuda.synchronize()
        E._lib.check(E._lib.lib().evs_check_index_errors(None))     # (a p2p wait that ran out of patience raises here)
        for k in range(steps):
            assert torch.equal(outs[k], wants[k][sl]), (policy, rank, k)
        # ... and the eager forward
        R = op.forward(x[sl], lo_, [idxs[0][t] for t in range(T)])
        op.p2p_flush()
        torch.cuda.synchronize()
        assert torch.equal(R, wants[0][sl]), (policy, rank, "forward")
        dist.barrier()
        for st in op._p2p.values():
            st.close()
        del op
        dist.barrier()
    dist.destroy_process_group()
    print("P2P_CHILD_OK rank %d" % rank)


if __name__ == "__main__":
    main()
