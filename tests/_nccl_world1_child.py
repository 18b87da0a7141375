"""Child process of tests/test_gpu_parity.py::test_sharded_step_through_rccl_at_world_1: the sharded op over HipBackend with the
REAL RCCL all_to_all_single (backend "nccl", world size 1 -- the one GPU a test box has): device buffers, split lists, the
asynchronous work handle and the planned / pipelined step, under every placement; results against the unsharded launch."""
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
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29571")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    rs = np.random.RandomState(3)
    ln = [5000, 7, 2600, 40, 9000, 3, 12000] + [100] * 19
    T, d, B = len(ln), 36, 1024 + 16
    ws = [torch.from_numpy(rs.uniform(-1, 1, size=(n, d)).astype(np.float32)).to(dev) for n in ln]
    ev = E.EVTables.from_fp32(ws)
    idx = torch.stack([torch.from_numpy(rs.randint(0, n, size=B)) for n in ln]).to(dev)
    off = torch.arange(B, device=dev).repeat(T, 1)
    x = torch.rand(B, d, device=dev)
    want = E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True)
    for policy in ("rows+replicate", "rows", "count", "rowsplit"):
        owner = sharded.plan_placement(ln, 1, policy, replicate_max_rows=2000)
        weights = {t: ws[t] for t in range(T)}
        op = sharded.ShardedEmbeddingInteract(ln, d, 0, 1, weights, sharded.HipBackend(dev), policy=policy, one_index_per_bag=True,
                                              replicate_max_rows=2000)
        op.force_exchange = True
        lo, li = [off[t] for t in range(T)], [idx[t] for t in range(T)]
        R = op.forward(x, lo, li)
        assert torch.equal(R, want), policy
        out = torch.empty_like(want)
        pl = op.plan(x, lo, li, out=out)
        for _ in range(3):
            op.step(pl)
        assert torch.equal(out, want), policy + " (planned)"
        h = op.run_start(pl)
        out.zero_()
        op.run_finish(pl, h)
        torch.cuda.synchronize()
        assert torch.equal(out, want), policy + " (pipelined)"
        assert not op.any_sharded or pl["recv"].data_ptr() != pl["send"].data_ptr(), "the exchange must have its own receive buffer"
        del op
    # round 5: the check that gates the device-to-device exchange (bench.py --exchange-mode auto, EVS_BENCH_P2P=1): one batch
    # through the RCCL collective and through exchange_mode "p2p", receive buffers bit-equal, the verdict agreed on over the group
    lo, li = [off[t] for t in range(T)], [idx[t] for t in range(T)]
    for policy in ("rows+replicate", "rowsplit"):
        assert sharded.verify_p2p_against_collective(ln, d, 0, 1, {t: ws[t] for t in range(T)}, sharded.HipBackend(dev), policy, lo, li,
                                                     force_exchange=True, one_index_per_bag=True, replicate_max_rows=2000), policy
    # round 6: exchange_mode "direct" -- the all-to-all issued by the extension itself (ONE ncclAllToAllv on the step's stream over
    # its own communicator; grouped ncclSend / ncclRecv as the other form): same receive buffers as all_to_all_single, same R
    from evstore_dlrm_amd import _ext
    X = _ext.ext()
    assert X is not None and X.rccl_available(), "the extension must reach RCCL on a GPU box"
    for use_v in ("1", "0"):
        os.environ["EVS_DIRECT_A2A_V"] = use_v
        sharded.direct_close()
        a2a = sharded.direct_comm(None, dev)
        assert a2a is not None and a2a.use_alltoallv == (use_v == "1")
        for policy in ("rows+replicate", "rows", "count", "rowsplit"):
            assert sharded.verify_p2p_against_collective(ln, d, 0, 1, {t: ws[t] for t in range(T)}, sharded.HipBackend(dev), policy, lo, li,
                                                         force_exchange=True, one_index_per_bag=True, replicate_max_rows=2000, other="direct"), policy
            op = sharded.ShardedEmbeddingInteract(ln, d, 0, 1, {t: ws[t] for t in range(T)}, sharded.HipBackend(dev), policy=policy,
                                                  one_index_per_bag=True, replicate_max_rows=2000)
            op.force_exchange = True
            op.exchange_mode = "direct"
            out = torch.empty_like(want)
            pl = op.plan(x, lo, li, out=out)
            for _ in range(3):
                op.step(pl)
            torch.cuda.synchronize()
            assert torch.equal(out, want), policy + " (direct)"
            assert not op.any_sharded or len(op._direct_plans) == 1, "one planned exchange, reused"
            # ... and the overlapped step: pool(i + 1) + its exchange on a side stream under the interaction of batch i, two pipeline
            # slots, event hand-overs -- eight batches in flight order, every R against the unsharded launch
            for how in (("events", "signals") if use_v == "1" else ()):
                op.overlap, op._ov = how, None
                xs2 = [torch.rand(B, d, device=dev) for _ in range(4)]
                idx2 = [torch.stack([torch.from_numpy(rs.randint(0, n, size=B)) for n in ln]).to(dev) for _ in range(4)]
                wants = [E.apply_emb_interact(xs2[j], off, idx2[j], ev, one_index_per_bag=True) for j in range(4)]
                outs = [torch.empty_like(want) for _ in range(2)]
                pls = {(j, sl): op.plan(xs2[j], lo, [idx2[j][t] for t in range(T)], out=outs[sl], slot=sl) for j in range(4) for sl in (0, 1)}
                got = []
                h = op.run_start(pls[(0, 0)])
                for i in range(8):
                    nxt = op.run_start(pls[((i + 1) % 4, (i + 1) % 2)]) if i + 1 < 8 else None
                    op.run_finish(pls[(i % 4, i % 2)], h)
                    got.append(outs[i % 2].clone())
                    h = nxt
                torch.cuda.synchronize()
                for i in range(8):
                    assert torch.equal(got[i], wants[i % 4]), "%s (overlapped step %d)" % (policy, i)
                assert not op.any_sharded or op._ov is not None
            del op
    sharded.direct_close()
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
