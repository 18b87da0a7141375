"""GPU: the exchange step of the table-sharded forward WITHOUT a collective call (sharded.py exchange_mode = "p2p",
csrc/evs_p2p.hip): the pooling kernel writes each peer's block straight into that peer's receive buffer, two flag words per
(peer, slot) hand the blocks over.  The layout contract is the collective's (extend_distributed.py:389-426, :444-465;
dlrm_s_pytorch.py:543-570): every test compares with what the all_to_all_single path / the single-process launch gives."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def E():
    import evstore_dlrm_amd as E
    assert torch.cuda.is_available()
    E._lib.lib()
    return E


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.mark.parametrize("bag1", [True, False])
@pytest.mark.parametrize("world,policy", [(2, "rows+replicate"), (4, "rows"), (8, "count"), (8, "rows+replicate"), (2, "rowsplit"), (4, "rowsplit"),
                                          (8, "rowsplit"), (1, "rows"), (1, "rowsplit")])
def test_p2p_virtual_ranks_equal_the_collective_layout(E, orc, world, policy, bag1):
    """`world` virtual ranks in one process, wired to each other's receive buffers by plain pointers: three lock-step rounds
    over the two pipeline slots (use counters 1, 1, 2: a slot is overwritten only after every consumer released it).  Every
    rank's R slice = the oracle on its batch slice, and -- one index per bag -- bit-equal to the same op through the
    by-hand all_to_all layout (what test_sharded_hip_world8_and_rowsplit_virtual_ranks pins)."""
    from evstore_dlrm_amd import sharded
    rs = np.random.RandomState(900 + world + len(policy))
    ln = [2, 2, 10131, 2202, 2, 2, 12, 2, 3, 93, 5, 8351, 3, 2, 14, 5461, 2, 5, 2, 4, 7046, 2, 2, 286, 2, 142]
    thresh, d, Bl = 2000, 36, 48
    Bg, T = world * Bl, len(ln)
    tabs = [rs.uniform(-1, 1, size=(n, d)).astype(np.float32) for n in ln]
    owner = sharded.plan_placement(ln, world, policy, replicate_max_rows=thresh)
    shared = {}

    def make_ops(mode):
        ops = []
        for r in range(world):
            held = {}
            for t in range(T):
                if owner[t] in (r, -1):
                    held[t] = torch.from_numpy(tabs[t])
                elif owner[t] == -2:
                    lo, hi = sharded.row_range(ln[t], r, world)
                    held[t] = torch.from_numpy(np.ascontiguousarray(tabs[t][lo:hi]))
            op = sharded.ShardedEmbeddingInteract(ln, d, r, world, held, sharded.HipBackend(torch.device("cuda")), policy=policy,
                                                  replicate_max_rows=thresh, one_index_per_bag=bag1)
            if mode == "p2p":
                op.exchange_mode = "p2p"
                op.p2p_virtual = shared
            ops.append(op)
        return ops

    ops, ref = make_ops("p2p"), make_ops("hand")
    for op in ops:
        if op.any_sharded:
            op._p2p_state(Bg)          # (the last one wires them all)
    for rnd in range(3):
        if bag1:
            idx = [rs.randint(0, n, size=Bg).astype(np.int64) for n in ln]
            off = [np.arange(Bg, dtype=np.int64) for _ in ln]
        else:
            lens = rs.randint(0, 4, size=(T, Bg))
            idx = [rs.randint(0, ln[k], size=lens[k].sum()).astype(np.int64) for k in range(T)]
            off = [np.concatenate([[0], np.cumsum(lens[k])[:-1]]).astype(np.int64) for k in range(T)]
        lS_i, lS_o = [torch.from_numpy(i).cuda() for i in idx], [torch.from_numpy(o).cuda() for o in off]
        x = torch.from_numpy(rs.uniform(-1, 1, size=(Bg, d)).astype(np.float32)).cuda()
        R_o = orc.interact_features(x.cpu().numpy(), orc.apply_emb(off, idx, tabs))
        slot = rnd % 2
        hs = [op.start(lS_o, lS_i, slot=slot) for op in ops]          # every rank pools straight into every peer's buffer
        for op in ops:
            op.p2p_flush()                                             # (one process: a queued signal would be waited for by the next launch)
        Rs = [op.finish(hs[r], x[r * Bl:(r + 1) * Bl], lS_o, lS_i) for r, op in enumerate(ops)]
        for op in ops:
            op.p2p_flush()
        assert E._lib.lib().evs_check_index_errors(None) == 0
        # the same ops through the collective's layout, assembled by hand
        sends = [op.pool(lS_o, lS_i)[0] if op.any_sharded else None for op in ref]
        for r, op in enumerate(ref):
            _, _, out_splits = op._splits(Bg)
            recv = torch.cat([sends[p].reshape(world, sends[p].numel() // world)[r] for p in range(world)]) if op.any_sharded else torch.empty(0, device="cuda")
            if ops[r].any_sharded:   # block for block what the peers wrote
                got = hs[r][1]
                assert got.numel() == recv.numel() and torch.equal(got, recv), (rnd, r)
            R_ref = op.finish((None, recv, Bg, Bl, out_splits), x[r * Bl:(r + 1) * Bl], lS_o, lS_i)
            assert torch.equal(Rs[r], R_ref), (rnd, r)
            np.testing.assert_allclose(Rs[r].cpu().numpy(), R_o[r * Bl:(r + 1) * Bl], rtol=RTOL, atol=2e-6 if bag1 else 1e-5)
    for op in ops:
        for st in op._p2p.values():
            st.close()


def test_p2p_exchange_between_processes():
    """Two processes, both on GPU 0, a gloo group for the handles: real hipIpcGetMemHandle / hipIpcOpenMemHandle mappings, the
    pool kernel of one process writing into the other's receive buffer, the flag hand-over across processes, seven pipelined
    steps over both slots under every placement -- each rank's slice bit-equal to the single-process fused launch
    (tests/_p2p_child.py)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_p2p_child.py")
    procs = [subprocess.Popen([sys.executable, child, str(r), "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, e = p.communicate()
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and "P2P_CHILD_OK" in o, (r, o[-1500:], e[-3000:])


def _run_children(extra, env_extra=None, timeout=420):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_p2p_child.py")
    procs = [subprocess.Popen([sys.executable, child, str(r), "2", str(port)] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, e = p.communicate()
        outs.append((p.returncode, o, e))
    return outs


def test_bench_side_line_between_processes():
    """bench.py's N > 1 side line (`exchange_p2p`: sharded.bench_p2p_side) with two processes on GPU 0 over a gloo group: the
    warm-up agreement, the settle loop and the timed loop over real IPC mappings; every rank returns a timing."""
    for r, (rc, o, e) in enumerate(_run_children(["bench"])):
        assert rc == 0 and "P2P_BENCH ok" in o, (r, o[-1500:], e[-3000:])


def test_bench_side_line_survives_a_rank_that_cannot_set_up():
    """A rank whose buffers cannot be allocated takes part in the hand-shake all the same: BOTH ranks come back with an error
    record (nobody waits for a peer that gave up), the process group stays usable (the barrier behind it passes)."""
    for r, (rc, o, e) in enumerate(_run_children(["bench"], {"EVS_P2P_INJECT_FAIL": "1"}, timeout=120)):
        assert rc == 0 and "P2P_BENCH error" in o and "injected" in o, (r, o[-1500:], e[-3000:])
