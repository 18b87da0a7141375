"""Table-sharded embedding lookup across the GPUs of one node (one process per GPU).

Reference behaviour (dlrm_s_pytorch.py:529-586 distributed_forward +
extend_distributed.py:389-465): tables are split over ranks, every rank pools ITS tables
for the FULL batch, one all-to-all turns "local tables x full batch" into "all tables x
local batch", interaction runs on the local batch slice.

MI355X-native re-design (same results, fewer bytes on xGMI and in HBM):
  * the gather kernel writes pooled vectors straight into the all-to-all send layout
    (B_global, T_own, d) -- no torch.cat;
  * ONE all_to_all_single over RCCL/xGMI (point-to-point links: each peer pair has its
    own link, so the exchange is one hop);
  * the receive blocks are consumed IN PLACE by the fused interaction kernel through
    per-feature (pointer, stride) pairs -- no split/view/cat, and any table placement
    (not only contiguous slices) keeps the reference's feature order;
  * placement policies: "count" (the reference's contiguous split, get_my_slice),
    "rows" (balance by row count, BASELINE.json north_star), and "rows+replicate":
    tables below `replicate_max_rows` are replicated on every rank and looked up
    locally inside the fused kernel for the local batch slice only, so their pooled
    vectors never cross xGMI (Kaggle: 21 of 26 tables, 82 MB; 5/26 of the bytes remain);
    "rowsplit": the tables above the threshold are split ROW-WISE into `world` contiguous
    ranges, the small ones replicated: with pooling factor 1 owning a
    whole table costs B_global lookups whatever its size (5 owners pool 131 072 rows each
    at world 8 while 3 ranks idle); row ranges make every rank pool ~ B_global * T_big / world
    rows.  A rank pools, for EVERY bag of the global batch, the partial sum over the rows of
    its range (evs_embedding_bag_sum_sharded) into its block of the same send buffer; the
    receiver reads the partial of the rank whose range holds the index (one index per bag:
    a select, bit-equal to the single-process row) or adds the `world` partials in rank
    order (multi-index bags: fp32 sums in another order than the single process).
The compute backend is injectable so the distributed plumbing is testable on CPU (gloo)
with the oracle standing in for the HIP kernels (tests/test_sharded_gloo.py).
"""
import ctypes as C
import time

import numpy as np
import os

import torch
import torch.distributed as dist

from . import _lib
from .dlrm_ops import EVTables, _stream_ptr


# ----------------------------------------------------------------------------- placement
def plan_placement(ln_emb, world, policy="rows+replicate", replicate_max_rows=1_000_000, replicate_budget_rows=None):
    """-> owner[t] in {0..world-1}, -1 (replicated on every rank) or -2 (split row-wise over all ranks).
    count            contiguous by table count (extend_distributed.get_my_slice)
    rowsplit         tables above replicate_max_rows split into `world` contiguous row ranges (rank r holds rows
                     [r*n//world, (r+1)*n//world)), the others on every rank
    rows             LPT greedy on row counts (BASELINE: "tables shard by row-count")
    rows+replicate   tables of at most replicate_max_rows rows on every rank, the rest by rows
    hbm              the same with the threshold set by memory: smallest tables first are replicated while their rows
                     fit replicate_budget_rows per GPU (288 GB of HBM3E: the whole Kaggle model is 4.9 GB), the rest
                     by rows.  Replicated tables need no exchange at all."""
    T = len(ln_emb)
    owner = [0] * T
    if policy == "hbm":
        budget = replicate_budget_rows if replicate_budget_rows is not None else sum(ln_emb)
        used = 0
        rep = set()
        for n, t in sorted((n, t) for t, n in enumerate(ln_emb)):   # smallest first, exactly while they fit
            if used + n > budget:
                break
            used += n
            rep.add(t)
        sub = [t for t in range(T) if t not in rep]
        sub_owner = plan_placement([ln_emb[t] for t in sub], world, "rows")
        owner = [-1] * T
        for t, o in zip(sub, sub_owner):
            owner[t] = o
        return owner
    if policy == "count":  # extend_distributed.get_my_slice: contiguous by table count
        k, m = divmod(T, world)
        t = 0
        for r in range(world):
            n = k + (1 if r < m else 0)
            for _ in range(n):
                owner[t] = r
                t += 1
        return owner
    if policy == "rowsplit":
        return [-2 if ln_emb[t] > replicate_max_rows else -1 for t in range(T)]
    if policy not in ("rows", "rows+replicate"):
        raise ValueError("unknown placement policy %r" % policy)
    shard = list(range(T))
    if policy == "rows+replicate":
        shard = [t for t in range(T) if ln_emb[t] > replicate_max_rows]
        for t in range(T):
            if ln_emb[t] <= replicate_max_rows:
                owner[t] = -1
    load = [0] * world
    for t in sorted(shard, key=lambda t: (-ln_emb[t], t)):  # LPT greedy on row counts
        r = min(range(world), key=lambda r: (load[r], r))
        owner[t] = r
        load[r] += ln_emb[t]
    return owner


def row_range(n, rank, world):
    """rows [lo, hi) of an n-row table that rank `rank` holds under the rowsplit placement"""
    return rank * n // world, (rank + 1) * n // world


def row_owner(idx, n, world):
    """rank whose range holds row idx (tensor or int): the largest r with r*n//world <= idx"""
    return ((idx + 1) * world + (n - 1)) // n - 1


# ----------------------------------------------------------------------------- backends
class HipBackend:
    """libevstore_hip.so kernels (the product path)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._cache = {}
        self.stream = None      # a raw HIP stream handle the launches go to instead of torch's current stream (the overlapped step)

    def _sp(self):
        return self.stream if self.stream is not None else _stream_ptr(self.device)

    def make_tables(self, weights, d):
        return EVTables([w.to(self.device) for w in weights], d, 32)

    def bag_sum_into(self, ev, table_ids_local, lS_o_rows, lS_i_rows, send, n_own, d, planned=False, bag1=False,
                     layout=None, row_lo=None, row_total=None, peer_delta=None):
        """pooled[b][j][:] for the j-th owned table, written into send (B, n_own, d).
        bag1: the caller states one index per bag (offsets = arange): the library's NULL-offsets row gather.
        planned: the caller passes the SAME list objects every step (plan / run_start): the pointer tables are then
        cached by list identity and the lists kept alive; one-off calls build them and keep nothing.
        layout = (B, float_offset, table_stride, bag_stride, peer_stride, bags_per_peer): write into the flat buffer `send`
        peer-major instead (evs_embedding_bag_sum_sharded); row_lo / row_total: the tables are row ranges of larger ones.
        peer_delta (p2p exchange): device int64 tensor, one entry per peer -- peer q's block goes to its local-layout address
        + peer_delta[q] floats, i.e. into that peer's receive buffer (evs_embedding_bag_sum_p2p)."""
        B = int(send.shape[0]) if layout is None else int(layout[0])
        n = len(table_ids_local)
        if n == 0 or B == 0:
            return
        # planned batches pass the same list objects every step: key on identity first (no per-step tuple building)
        fast = ("gid", id(lS_i_rows), id(lS_o_rows), send.data_ptr(), bag1, layout, None if peer_delta is None else peer_delta.data_ptr())
        ent = self._cache.get(fast) if planned else None
        if ent is None:
            # "one index per bag" is a statement about the batch: honour it only when every index list HAS B entries
            # (as apply_emb does); anything else pools through the offsets the caller gave
            bag1 = bag1 and all(int(t.numel()) == B for t in lS_i_rows)
            ent = ((C.c_void_p * n)(*[ev._tables_c[k] for k in table_ids_local]),
                   (C.c_int64 * n)(*[ev.n_rows[k] for k in table_ids_local]),
                   (C.c_void_p * n)(*[t.data_ptr() for t in lS_i_rows]),
                   None if bag1 else (C.c_void_p * n)(*[t.data_ptr() for t in lS_o_rows]),
                   (C.c_int64 * n)(*[int(t.numel()) for t in lS_i_rows]),
                   lS_i_rows, lS_o_rows,  # the lists are kept alive so their ids stay unique
                   None if row_lo is None else (C.c_int64 * n)(*row_lo),
                   None if row_total is None else (C.c_int64 * n)(*row_total))
            if planned:
                if len(self._cache) > 256:
                    self._cache.clear()
                self._cache[fast] = ent
        L = _lib.lib()
        if layout is None and row_lo is None and peer_delta is None:
            _lib.check(L.evs_embedding_bag_sum(n, B, d, ev.codec, ent[0], ent[1], ent[2], ent[3], ent[4], None,
                                               send.data_ptr(), d, n_own * d, self._sp()))
            return
        _, foff, tstride, bstride, pstride, bpp = layout if layout is not None else (B, 0, d, n_own * d, 0, 0)
        if peer_delta is not None and bpp < B:
            _lib.check(L.evs_embedding_bag_sum_p2p(n, B, d, ev.codec, ent[0], ent[1], ent[7], ent[8], ent[2], ent[3], ent[4], None,
                                                   send.data_ptr() + 4 * foff, tstride, bstride, pstride, bpp, peer_delta.data_ptr(),
                                                   self._sp()))
            return
        _lib.check(L.evs_embedding_bag_sum_sharded(n, B, d, ev.codec, ent[0], ent[1], ent[7], ent[8], ent[2], ent[3], ent[4], None,
                                                   send.data_ptr() + 4 * foff, tstride, bstride, pstride, bpp,
                                                   self._sp()))

    def route(self, src_rows, n_rows, world, row_off, dst_rows, key=None):
        """row-split tables, receiver side: dst[k][b] = row_off[owner(src[k][b])] + b in ONE launch (evs_rowsplit_route)"""
        n = len(src_rows)
        ent = self._cache.get(key) if key is not None else None
        if ent is None:
            ent = ((C.c_void_p * n)(*[t.data_ptr() for t in src_rows]), (C.c_int64 * n)(*n_rows), (C.c_int64 * world)(*row_off),
                   (C.c_void_p * n)(*[t.data_ptr() for t in dst_rows]), src_rows, dst_rows)
            if key is not None:
                if len(self._cache) > 256:
                    self._cache.clear()
                self._cache[key] = ent
        _lib.check(_lib.lib().evs_rowsplit_route(n, int(src_rows[0].numel()), world, ent[0], ent[1], ent[2], ent[3],
                                                 self._sp()))

    def interact_mixed(self, x, specs, ev, d, itself, out=None, planned=False):
        """specs[t]: ("dense", tensor(B,d) view) | ("indirect", local_table_id, idx, off, nnz, off_len) |
        ("gathered", rows, n_rows, idx, off, nnz, off_len): like "indirect" over the fp32 rows of `rows` (a flat tensor:
        row r = rows[r*d : (r+1)*d]) -- the row-split tables' partials inside the receive buffer."""
        B = int(x.shape[0])
        F = len(specs) + 1
        P = F * (F + 1) // 2 if itself else F * (F - 1) // 2
        R = out if out is not None else torch.empty((B, d + P), dtype=torch.float32, device=self.device)
        fast = ("fid", id(specs), x.data_ptr(), R.data_ptr())   # a planned batch re-uses the same specs list
        ent = self._cache.get(fast) if planned else None
        if ent is None:
            feats = (_lib.EvsFeature * F)()
            feats[0].src, feats[0].stride = x.data_ptr(), int(x.stride(0)) if B > 1 else d
            for t, s in enumerate(specs):
                f = feats[t + 1]
                if s[0] == "dense":
                    v = s[1]
                    f.src, f.stride = v.data_ptr(), int(v.stride(0)) if B > 1 else d
                elif s[0] == "gathered":
                    _, rows, n_rows, idx, off, nnz, off_len = s
                    f.src, f.indices = rows.data_ptr(), idx.data_ptr()
                    f.offsets = off.data_ptr() if off is not None else None
                    f.nnz, f.n_rows, f.offsets_len = int(nnz), int(n_rows), int(off_len)
                else:
                    _, k, idx, off, nnz, off_len = s
                    # empty index tensor: data_ptr() is NULL, which the C ABI reads as "dense" -- pass the table address
                    f.src, f.indices = ev._tables_c[k], (idx.data_ptr() or ev._tables_c[k])
                    f.offsets = off.data_ptr() if off is not None else None
                    f.nnz, f.n_rows, f.offsets_len = int(nnz), ev.n_rows[k], int(off_len)
            ent = (feats, specs if planned else None)   # keep the list alive only when it can recur
            if planned:
                if len(self._cache) > 256:
                    self._cache.clear()
                self._cache[fast] = ent
        _lib.check(_lib.lib().evs_emb_interact_dot(B, F, d, ev.codec if ev is not None else 32, ent[0],
                                                   int(bool(itself)), R.data_ptr(), self._sp()))
        return R


# ----------------------------------------------------------------------------- exchange without a collective call
class _P2PExchange:
    """exchange_mode = "p2p" for one global batch size: per pipeline slot a receive buffer and a block of flag words in
    fine-grained device memory, IPC handles exchanged ONCE over the process group, every peer's buffers mapped here.  The
    pooling kernel writes each peer's block straight into that peer's receive buffer (HipBackend.bag_sum_into(...,
    peer_delta=...)); the hand-over is ready[src][slot] = k ("use k of your slot holds my block") and free[dst][slot] = k
    ("I am through with use k of it"), written with evs_p2p_sync launches on the step's own stream (csrc/evs_p2p.hip).  Signals
    are queued and ride in the NEXT sync launch (two tiny launches per step instead of four); flush() sends what is queued.
    Layout contract: what all_to_all_single(recv, send, out_splits, in_splits) would have delivered, block for block
    (extend_distributed.py:389-426, :444-465)."""

    def __init__(self, op, Bg, n_slots=2, virtual_peers=None):
        """virtual_peers: None = one process per rank (handles over op.group); else the list that will hold every virtual
        rank's exchange object of ONE process (tests): the last one to be made connects them all by plain pointers"""
        L = _lib.lib()
        self.op, self.W, self.r, self.n_slots = op, op.world, op.rank, n_slots
        Bl, in_splits, out_splits = op._splits(Bg)
        self.blk = in_splits[0]                      # floats this rank sends to every peer
        self.my_off = sum(out_splits[:op.rank])      # where this rank's block starts inside ANY peer's receive buffer
        self.n_recv = sum(out_splits)
        self.dev = op.backend.device
        self.own_recv, self.own_flags, self.recv_t, handles = [], [], [], []
        self._opened = []
        real = self.W > 1 and virtual_peers is None      # one process per rank: every step below is agreed on by all ranks
        err = None
        try:
            if os.environ.get("EVS_P2P_INJECT_FAIL") == str(self.r):   # (tests: a rank whose set-up fails must not strand the others)
                raise RuntimeError("injected set-up failure")
            with torch.cuda.device(self.dev):
                for s_ in range(n_slots):
                    pr, pf = C.c_void_p(), C.c_void_p()
                    _lib.check(L.evs_p2p_alloc(C.byref(pr), max(4 * self.n_recv, 64)))
                    self.own_recv.append(pr.value)
                    _lib.check(L.evs_p2p_alloc(C.byref(pf), 4 * 2 * max(self.W, 1) + 64))
                    self.own_flags.append(pf.value)
                    hr, hf = (C.c_char * 64)(), (C.c_char * 64)()
                    if real:
                        _lib.check(L.evs_p2p_ipc_export(pr, hr)); _lib.check(L.evs_p2p_ipc_export(pf, hf))
                    handles.append((bytes(hr), bytes(hf)))
        except Exception as ex:
            if not real:
                self.close()
                raise
            err = repr(ex)
        if real:   # (a rank that failed still takes part in the collective: nobody is left waiting for it)
            allh = [None] * self.W
            dist.all_gather_object(allh, (err, handles), group=op.group)
            bad = ["rank %d: %s" % (p, a[0]) for p, a in enumerate(allh) if a[0]]
            if bad:
                self.close()
                raise RuntimeError("p2p exchange: buffer allocation / export failed (%s)" % "; ".join(bad))
        # the receive buffer as a tensor the interaction kernel's feature specs can view (no copy: __cuda_array_interface__)
        for s_ in range(n_slots):
            self.recv_t.append(_tensor_over(self.own_recv[s_], self.n_recv, self.dev, self))
        self.peer_recv = [[None] * self.W for _ in range(n_slots)]
        self.peer_flags = [[None] * self.W for _ in range(n_slots)]
        self.k_pool = [0] * n_slots       # uses of a slot this rank has pooled into / consumed
        self.k_done = [0] * n_slots
        self._pending = None              # (pointer array, value): a signal that rides in the next sync launch
        self._virtual = virtual_peers is not None
        if self._virtual:
            virtual_peers[self.r] = self
            if all(v is not None for v in virtual_peers):
                for v in virtual_peers:
                    for s_ in range(n_slots):
                        for p in range(self.W):
                            v.peer_recv[s_][p], v.peer_flags[s_][p] = virtual_peers[p].own_recv[s_], virtual_peers[p].own_flags[s_]
                    v._wire()
            return
        try:
            for s_ in range(n_slots):
                for p in range(self.W):
                    if p == self.r:
                        self.peer_recv[s_][p], self.peer_flags[s_][p] = self.own_recv[s_], self.own_flags[s_]
                        continue
                    with torch.cuda.device(self.dev):
                        for which, h in enumerate(allh[p][1][s_]):
                            out = C.c_void_p()
                            _lib.check(L.evs_p2p_ipc_open(C.create_string_buffer(h, 64), C.byref(out)))
                            self._opened.append(out.value)
                            if which == 0:
                                self.peer_recv[s_][p] = out.value
                            else:
                                self.peer_flags[s_][p] = out.value
        except Exception as ex:
            if not real:
                self.close()
                raise
            err = repr(ex)
        if real:
            oks = [None] * self.W
            dist.all_gather_object(oks, err, group=op.group)   # (also: everybody has mapped everybody before the first write)
            bad = ["rank %d: %s" % (p, e) for p, e in enumerate(oks) if e]
            if bad:
                self.close()
                raise RuntimeError("p2p exchange: mapping a peer's buffers failed (%s)" % "; ".join(bad))
        self._wire()

    def _wire(self):
        n_slots = self.n_slots
        # peer q's block: local-layout address (own receive buffer + my_off + q * blk) -> peer q's receive buffer + my_off
        self.delta, self.out_t = [], []
        for s_ in range(n_slots):
            base = self.own_recv[s_] + 4 * self.my_off
            dl = [((self.peer_recv[s_][q] + 4 * self.my_off) - (base + 4 * q * self.blk)) // 4 for q in range(self.W)]
            self.delta.append(torch.tensor(dl, dtype=torch.int64, device=self.dev))
            self.out_t.append(self.recv_t[s_][self.my_off:])       # the pool kernels' `out`: this rank's block inside its own buffer
        # flag words: ready[W] then free[W] in every rank's block
        self.sig_ready = [(C.c_void_p * self.W)(*[self.peer_flags[s_][p] + 4 * self.r for p in range(self.W)]) for s_ in range(n_slots)]
        self.sig_free = [(C.c_void_p * self.W)(*[self.peer_flags[s_][p] + 4 * (self.W + self.r) for p in range(self.W)]) for s_ in range(n_slots)]
        self.wait_ready = [(C.c_void_p * self.W)(*[self.own_flags[s_] + 4 * q for q in range(self.W)]) for s_ in range(n_slots)]
        self.wait_free = [(C.c_void_p * self.W)(*[self.own_flags[s_] + 4 * (self.W + q) for q in range(self.W)]) for s_ in range(n_slots)]

    def _sync(self, wait=None, wait_value=0):
        sig, sv = self._pending if self._pending is not None else (None, 0)
        self._pending = None
        _lib.check(_lib.lib().evs_p2p_sync(self.W if sig is not None else 0, sig, sv, self.W if wait is not None else 0, wait, wait_value,
                                           _stream_ptr(self.dev)))

    def flush(self):
        if self._pending is not None:
            self._sync()

    def begin_pool(self, slot):
        """before the pooling launch of the slot's next use: every peer is through with the previous one"""
        self.k_pool[slot] += 1
        self._sync(self.wait_free[slot], (self.k_pool[slot] - 1) & 0xffffffff)
        return self.k_pool[slot]

    def end_pool(self, slot):
        if self._pending is not None:
            self._sync()
        self._pending = (self.sig_ready[slot], self.k_pool[slot] & 0xffffffff)

    def begin_consume(self, slot):
        self.k_done[slot] += 1
        self._sync(self.wait_ready[slot], self.k_done[slot] & 0xffffffff)

    def end_consume(self, slot):
        if self._pending is not None:
            self._sync()
        self._pending = (self.sig_free[slot], self.k_done[slot] & 0xffffffff)

    def __del__(self):   # (an op dropped without close(): its buffers and mappings go with it; peers are past their last step by then)
        try:
            if self._opened or self.own_recv or self.own_flags:
                self.close()
        except Exception:
            pass

    def close(self):
        L = _lib.lib()
        try:
            torch.cuda.synchronize(self.dev)
        except Exception:
            pass
        for p in self._opened:
            L.evs_p2p_ipc_close(C.c_void_p(p))
        for p in self.own_recv + self.own_flags:
            if p:
                L.evs_p2p_free(C.c_void_p(p))
        self._opened, self.own_recv, self.own_flags = [], [], []


class _RawMem:
    """a device allocation of the library as something torch.as_tensor can view (kept alive by `owner`)"""
    def __init__(self, ptr, n_floats, owner):
        self.__cuda_array_interface__ = {"shape": (max(int(n_floats), 1),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
        self._owner = owner


def _tensor_over(ptr, n_floats, device, owner):
    t = torch.as_tensor(_RawMem(ptr, n_floats, owner), device=device)
    return t[:n_floats]


# ----------------------------------------------------------------------------- the sharded op
class ShardedEmbeddingInteract:
    """apply_emb + all-to-all + interact_features for one rank.

    local_weights: dict {global table id -> (n,d) fp32 tensor} for the tables this rank
    holds (owned + replicated).  Every rank must use the same ln_emb / policy.
    """

    def __init__(self, ln_emb, d, rank, world, local_weights, backend, policy="rows+replicate",
                 replicate_max_rows=1_000_000, group=None, itself=False, one_index_per_bag=False,
                 replicate_budget_rows=None):
        self.ln_emb, self.d, self.rank, self.world = list(ln_emb), int(d), rank, world
        self.group, self.itself, self.backend = group, itself, backend
        # lS_o[k] == arange(B) for every table (Criteo collate): replicated tables skip the offsets stage
        self.one_index_per_bag = one_index_per_bag
        self.owner = plan_placement(ln_emb, world, policy, replicate_max_rows, replicate_budget_rows)
        self.own = [[t for t in range(len(ln_emb)) if self.owner[t] == r] for r in range(world)]
        self.my_own = self.own[rank]
        self.replicated = [t for t in range(len(ln_emb)) if self.owner[t] == -1]
        self.split = [t for t in range(len(ln_emb)) if self.owner[t] == -2]   # row-split: this rank holds row_range(n, rank, world)
        self.any_sharded = any(o >= 0 or o == -2 for o in self.owner)
        held = self.my_own + self.replicated + self.split
        assert sorted(local_weights.keys()) == sorted(held), "rank %d must hold tables %s" % (rank, held)
        for t in self.split:
            lo, hi = row_range(self.ln_emb[t], rank, world)
            assert int(local_weights[t].shape[0]) == hi - lo, "rank %d holds rows [%d, %d) of table %d" % (rank, lo, hi, t)
        self.local_id = {t: i for i, t in enumerate(held)}
        self.ev = backend.make_tables([local_weights[t] for t in held], d)
        self._bufs = {}
        self._route = {}
        self._p2p = {}
        self._direct_plans = {}

    def tables_held(self):
        return self.my_own + self.replicated + self.split

    # a2a element counts for a global batch of Bg samples.  The block a rank sends to every peer: its owned tables'
    # pooled vectors for the peer's samples, sample-major (Bl, n_own, d), then its partials of the row-split tables,
    # table-major (n_split, Bl, d) -- the second section has the same size on every rank
    def _splits(self, Bg):
        assert Bg % self.world == 0, "batch_size %d can not split across %d ranks evenly" % (Bg, self.world)
        Bl = Bg // self.world
        S = len(self.split) * Bl * self.d
        in_splits = [Bl * len(self.my_own) * self.d + S] * self.world
        out_splits = [Bl * len(self.own[p]) * self.d + S for p in range(self.world)]
        return Bl, in_splits, out_splits

    p2p_virtual = None    # tests: {Bg: [None] * world} shared by the virtual ranks of one process (see _P2PExchange)

    def _p2p_state(self, Bg):
        st = self._p2p.get(Bg)
        if st is None:
            vp = None if self.p2p_virtual is None else self.p2p_virtual.setdefault(Bg, [None] * self.world)
            st = self._p2p[Bg] = _P2PExchange(self, Bg, virtual_peers=vp)
        return st

    def _p2p_slot_of(self, recv, Bg):
        st = self._p2p_state(Bg)
        for s_ in range(st.n_slots):
            if st.recv_t[s_].data_ptr() == recv.data_ptr():
                return st, s_
        return st, None   # (a receive buffer the caller assembled: virtual-rank tests of the collective layout)

    def p2p_flush(self):
        """send the hand-over signals still queued behind the last step (exchange_mode "p2p"; a no-op otherwise)"""
        for st in self._p2p.values():
            st.flush()

    def _buffers(self, Bg, slot, like):
        if self.exchange_mode == "p2p" and self.any_sharded:
            # p2p: `send` is this rank's block inside its OWN receive buffer (the pool kernels' reference address; every peer's
            # block is redirected into that peer's buffer by the per-peer delta), `recv` the whole receive buffer of the slot
            st = self._p2p_state(Bg)
            return st.out_t[slot % st.n_slots], st.recv_t[slot % st.n_slots]
        key = (Bg, slot)
        if key not in self._bufs:
            Bl, in_splits, out_splits = self._splits(Bg)
            if self.split:
                send = like.new_empty((sum(in_splits),), dtype=torch.float32)
            else:
                send = like.new_empty((Bg, max(len(self.my_own), 0), self.d), dtype=torch.float32)
            # one rank: nothing to exchange -- the "received" block is the send buffer itself (no copy)
            recv = send.view(-1) if not self._exchanges() else like.new_empty((sum(out_splits),), dtype=torch.float32)
            self._bufs[key] = (send, recv)
        return self._bufs[key]

    def _pool_into(self, send, Bg, lo_rows, li_rows, planned=False, lo_split=None, li_split=None):
        """the pooling launch(es) of one batch: owned tables, then this rank's row ranges of the split tables"""
        d, n_own = self.d, len(self.my_own)
        Bl = Bg // self.world
        if self.exchange_mode == "p2p":
            st = self._p2p_state(Bg)
            slot = self._p2p_slot
            blk = st.blk
            if not self._p2p_begun:
                st.begin_pool(slot)
            self._p2p_begun = False
            pd = st.delta[slot] if self.world > 1 else None    # (one rank: its own buffer, the plain layout)
            if n_own:
                self.backend.bag_sum_into(self.ev, [self.local_id[t] for t in self.my_own], lo_rows, li_rows, send, n_own, d,
                                          planned=planned, bag1=self.one_index_per_bag, layout=(Bg, 0, d, n_own * d, blk, Bl), peer_delta=pd)
            if self.split:
                self.backend.bag_sum_into(self.ev, [self.local_id[t] for t in self.split], lo_split, li_split, send, len(self.split), d,
                                          planned=planned, bag1=self.one_index_per_bag,
                                          layout=(Bg, Bl * n_own * d, Bl * d, d, blk, Bl),
                                          row_lo=[row_range(self.ln_emb[t], self.rank, self.world)[0] for t in self.split],
                                          row_total=[self.ln_emb[t] for t in self.split], peer_delta=pd)
            st.end_pool(slot)
            return
        if not self.split:
            self.backend.bag_sum_into(self.ev, [self.local_id[t] for t in self.my_own], lo_rows, li_rows, send, n_own, d,
                                      planned=planned, bag1=self.one_index_per_bag)
            return
        blk = Bl * n_own * d + len(self.split) * Bl * d
        if n_own:
            self.backend.bag_sum_into(self.ev, [self.local_id[t] for t in self.my_own], lo_rows, li_rows, send, n_own, d,
                                      planned=planned, bag1=self.one_index_per_bag, layout=(Bg, 0, d, n_own * d, blk, Bl))
        self.backend.bag_sum_into(self.ev, [self.local_id[t] for t in self.split], lo_split, li_split, send, len(self.split), d,
                                  planned=planned, bag1=self.one_index_per_bag,
                                  layout=(Bg, Bl * n_own * d, Bl * d, d, blk, Bl),
                                  row_lo=[row_range(self.ln_emb[t], self.rank, self.world)[0] for t in self.split],
                                  row_total=[self.ln_emb[t] for t in self.split])

    # one rank has nothing to exchange and the received block is the send buffer itself -- unless force_exchange asks for
    # the collective anyway (tests: the RCCL call with this op's buffers and split lists, on the one GPU a box has)
    force_exchange = False
    _p2p_slot = 0
    _p2p_begun = False

    def _exchanges(self):
        return (self.world > 1 or self.force_exchange) and self.exchange_mode != "p2p"   # (p2p: the pool launch IS the exchange)

    def pool(self, lS_o, lS_i, slot=0):
        """Pool the owned tables (and this rank's row ranges of the split tables) for the full batch into the send layout."""
        Bg = int(lS_o[0].shape[0])
        like = lS_o[0].new_empty((0,), dtype=torch.float32)
        send, recv = self._buffers(Bg, slot, like)
        self._p2p_slot = slot % 2
        self._pool_into(send, Bg, [lS_o[t] for t in self.my_own], [lS_i[t] for t in self.my_own],
                        lo_split=[lS_o[t] for t in self.split], li_split=[lS_i[t] for t in self.split])
        return send, recv

    def start(self, lS_o, lS_i, slot=0):
        """pool() + launch the all-to-all (async)."""
        Bg = int(lS_o[0].shape[0])
        Bl, in_splits, out_splits = self._splits(Bg)
        if not self.any_sharded:   # every table replicated: nothing to pool for others, nothing to exchange
            like = lS_o[0].new_empty((0,), dtype=torch.float32)
            return (None, self._buffers(Bg, slot, like)[1], Bg, Bl, out_splits)
        send, recv = self.pool(lS_o, lS_i, slot)
        work = None
        if self._exchanges():
            work = self._exchange(recv, send.view(-1), out_splits, in_splits)
        return (work, recv, Bg, Bl, out_splits)

    # ---- where the receiver finds things ------------------------------------------------------------------------------
    def _split_geometry(self, Bg):
        """(float offset of block p's split section in the receive buffer, row offset of block p relative to block 0)"""
        Bl, _, out_splits = self._splits(Bg)
        soff, pos = [], 0
        for p in range(self.world):
            soff.append(pos + Bl * len(self.own[p]) * self.d)
            pos += out_splits[p]
        return soff, [(o - soff[0]) // self.d for o in soff]

    def _split_specs(self, recv, Bg, lS_o, lS_i):
        """one "gathered" feature per row-split table: its rows are the `world` partials inside the receive buffer.
        One index per bag: sample b reads the partial of the rank whose range holds idx[b] (a select: the row itself).
        Otherwise: the bag of sample b is its `world` partials, added in rank order."""
        if not self.split:
            return {}
        Bl = Bg // self.world
        b0 = self.rank * Bl
        soff, rowoff = self._split_geometry(Bg)
        n_virtual = rowoff[-1] + Bl
        dev = recv.device
        key = (Bg, str(dev))
        if key not in self._route:
            ro = torch.tensor(rowoff, dtype=torch.int64, device=dev)
            ar = torch.arange(Bl, dtype=torch.int64, device=dev)
            self._route[key] = (ro, ar, (ar[:, None] + ro[None, :]).reshape(-1).contiguous(), ar * self.world,
                                torch.tensor([self.ln_emb[t] for t in self.split], dtype=torch.int64, device=dev)[:, None])
        ro, ar, all_partials, bag_starts, n_col = self._route[key]
        out = {}
        if self.one_index_per_bag:
            v = torch.stack([lS_i[t][b0:b0 + Bl] for t in self.split])                  # (n_split, Bl)
            own = row_owner(v, n_col, self.world).clamp_(0, self.world - 1)              # (a bad index: the pool flagged it)
            idx2 = ro[own] + ar[None, :]
        for j, t in enumerate(self.split):
            rows = recv[soff[0] + j * Bl * self.d:]
            if self.one_index_per_bag:
                out[t] = ("gathered", rows, n_virtual, idx2[j], None, Bl, 0)
            else:
                out[t] = ("gathered", rows, n_virtual, all_partials, bag_starts, Bl * self.world, Bl)
        return out

    def _specs(self, recv, Bg, Bl, out_splits, lS_o, lS_i):
        b0 = self.rank * Bl
        specs = [None] * len(self.ln_emb)
        pos = 0
        for p in range(self.world):
            n = len(self.own[p])
            if n:
                block = recv[pos:pos + Bl * n * self.d].view(Bl, n, self.d)
                for j, t in enumerate(self.own[p]):
                    specs[t] = ("dense", block[:, j, :])
            pos += out_splits[p]
        for t in self.replicated:  # looked up locally, only for this rank's samples
            if self.one_index_per_bag:
                specs[t] = ("indirect", self.local_id[t], lS_i[t][b0:], None, Bg - b0, 0)
            else:
                specs[t] = ("indirect", self.local_id[t], lS_i[t], lS_o[t][b0:], int(lS_i[t].numel()), Bg - b0)
        for t, sp in self._split_specs(recv, Bg, lS_o, lS_i).items():
            specs[t] = sp
        return specs

    def finish(self, handle, x_local, lS_o, lS_i, out=None):
        """Wait for the exchange, then R = interact_features(x_local, ly) on this rank's batch slice."""
        work, recv, Bg, Bl, out_splits = handle
        if work is not None:
            work.wait()
        st, slot = self._p2p_slot_of(recv, Bg) if (self.exchange_mode == "p2p" and self.any_sharded) else (None, None)
        if slot is not None:
            st.begin_consume(slot)       # every source's block of this use has arrived
        R = self.backend.interact_mixed(x_local, self._specs(recv, Bg, Bl, out_splits, lS_o, lS_i), self.ev, self.d,
                                        self.itself, out=out)
        if slot is not None:
            st.end_consume(slot)         # (queued: rides in the next sync launch; p2p_flush() sends it now)
        return R

    # ---- pre-planned steady state: all per-batch Python (views, pointer tables) done once -------------
    def plan(self, x_local, lS_o, lS_i, out=None, slot=0):
        """Prepare a batch whose tensors will be reused (serving loops re-use staged buffers): returns an
        object for run_start / run_finish that only launches kernels and the collective."""
        Bg = int(lS_o[0].shape[0])
        Bl, in_splits, out_splits = self._splits(Bg)
        like = lS_o[0].new_empty((0,), dtype=torch.float32)
        send, recv = self._buffers(Bg, slot, like)
        specs = self._specs(recv, Bg, Bl, out_splits, lS_o, lS_i)
        pl = {"send": send, "recv": recv, "in": in_splits, "out": out_splits, "specs": specs, "x": x_local, "slot": slot % 2,
              "R": out, "Bg": Bg, "ids": [self.local_id[t] for t in self.my_own],
              "lo": [lS_o[t] for t in self.my_own], "li": [lS_i[t] for t in self.my_own],
              "lo_split": [lS_o[t] for t in self.split], "li_split": [lS_i[t] for t in self.split]}
        if self.split and self.one_index_per_bag:
            # the route of a row-split lookup depends on the batch's indices: run_finish recomputes it INTO these tensors
            # every step (a serving loop refills its staged index buffers in place)
            pl["route_dst"] = [specs[t][3] for t in self.split]
            pl["route_src"] = [lS_i[t][self.rank * Bl:(self.rank + 1) * Bl] for t in self.split]
        return pl

    def _reroute(self, pl):
        Bg = pl["Bg"]
        Bl = Bg // self.world
        if hasattr(self.backend, "route"):   # one launch
            _, rowoff = self._split_geometry(Bg)
            self.backend.route(pl["route_src"], [self.ln_emb[t] for t in self.split], self.world, rowoff, pl["route_dst"],
                               key=("rid", id(pl["route_src"]), id(pl["route_dst"])))
            return
        ro, ar, _, _, n_col = self._route[(Bg, str(pl["recv"].device))]
        own = row_owner(torch.stack(pl["route_src"]), n_col, self.world).clamp_(0, self.world - 1)
        idx2 = ro[own] + ar[None, :]
        for j, dst in enumerate(pl["route_dst"]):
            dst.copy_(idx2[j])

    def run_start(self, pl):
        """Pooling gather of the owned tables + the all-to-all (async); nothing when every table is replicated.  Same stream as the interaction: a
        separate pooling stream with per-slot events was measured at world=1 and LOST (66-76 us per step against
        47) -- at this batch size the step is bounded by host-side launch cost, not by GPU overlap."""
        if not self.any_sharded:
            return None
        tr = self.trace
        if tr is not None:
            tr["n"] += 1
            if tr["n"] % tr["every"]:
                tr = None
        self._p2p_slot = pl["slot"]
        if self.exchange_mode == "p2p":   # the wait for the peers' release in front of the stamped interval, not inside it
            self._p2p_state(pl["Bg"]).begin_pool(pl["slot"])
            self._p2p_begun = True
        ov = self._overlap_state() if (self.overlap and self.exchange_mode in ("direct", "inline") and self.trace is None) else None
        if ov is not None and not (self._exchanges() and self.exchange_mode == "inline"):
            # round 6, the reference's overlap (dlrm_s_pytorch.py:564-569 hides its all-to-all under the bottom MLP): pool(i + 1) and
            # its exchange on a stream of their own under the interaction of batch i.  Two hand-overs per step: this slot's
            # buffers are free once the interaction that read them last has run; the interaction waits for this exchange.
            side, slot = ov["side"], pl["slot"]
            L = _lib.lib()
            if ov["sig"] is not None:   # hand-overs as signal words of the command processors (evs_stream_wait_value: >=)
                if ov["n_free"][slot]:
                    _lib.check(L.evs_stream_wait_value(side.cuda_stream, ov["sig"][slot][1], ov["n_free"][slot]))
            else:
                side.wait_event(ov["free"][slot])
            self.backend.stream = side.cuda_stream
            try:
                self._pool_into(pl["send"], pl["Bg"], pl["lo"], pl["li"], planned=True, lo_split=pl["lo_split"], li_split=pl["li_split"])
                if self._exchanges():
                    with torch.cuda.stream(side):   # (DirectA2A.run takes torch's current stream)
                        self._exchange(pl["recv"], pl["send"].view(-1), pl["out"], pl["in"])
            finally:
                self.backend.stream = None
            if ov["sig"] is not None:
                ov["n_ready"][slot] += 1
                _lib.check(L.evs_stream_write_value(side.cuda_stream, ov["sig"][slot][0], ov["n_ready"][slot]))
            else:
                ov["ready"][slot].record(side)
            return ("overlap", slot)
        if tr is not None:
            tr["pool"].append(self._stamp())
        self._pool_into(pl["send"], pl["Bg"], pl["lo"], pl["li"], planned=True, lo_split=pl["lo_split"], li_split=pl["li_split"])
        if tr is not None:
            tr["pool"].append(self._stamp())
        if self._exchanges():
            return self._exchange(pl["recv"], pl["send"].view(-1), pl["out"], pl["in"])
        return None   # one rank: recv aliases send

    # How the collective is issued.  The call path alone, measured at world size 1 (tools/a2a_cost.py: a self-exchange, no
    # link time): all_to_all_single(async_op=True) + wait() costs 37.8 us of host time and 39.6 us of stream time per call (a
    # Work object, an extra pair of cross-stream event waits); async_op=False 14.1 us host, 24.1 us stream.
    #   "inline" (default): async_op=False, everything in stream order -- pool(i + 1), exchange(i + 1), interaction(i); the
    #             two-deep pipeline overlaps HOST work (the next batch's launches) with device work, not the exchange with
    #             the interaction;
    #   "async":  async_op=True, the handle waited on in front of the interaction: the exchange of batch i + 1 runs on
    #             RCCL's stream under the interaction of batch i -- at the price of the dearer call.
    # One rank with the exchange forced (bench.py --force-sharded --force-exchange): 52-58 us per step inline, 63 us async.
    # Round 5: pool(i + 1) + the collective on a stream of their own under the interaction of batch i, the two hand-overs per
    # step as signal words (evs_stream_write_value / _wait_value) instead of events, was built and measured: 69-75 us against
    # 52-57 inline on the same box -- the step is bound by the HOST cost of the collective's call path (14-38 us per call),
    # and two more stream operations per step add to it; what takes the collective off the step is exchange_mode "p2p"
    # (29.5 us; picked by "auto" -- bench.py -- once one batch through both exchanges gave bit-equal receive buffers).
    #   "direct" (round 6): the same collective issued by the extension itself -- ONE ncclAllToAllv on the step's stream over a
    #             communicator of the extension's own (csrc/evs_torch_ext.cpp: DirectA2A), no ProcessGroup call in the step;
    #             falls back to "inline" where the extension or RCCL is missing (direct_comm returns None).
    exchange_mode = "inline"

    def _exchange(self, recv, send, out_splits, in_splits):
        if self.exchange_mode == "direct":
            a2a = direct_comm(self.group, recv.device)
            if a2a is not None:
                key = (recv.data_ptr(), send.data_ptr(), tuple(out_splits), tuple(in_splits))
                k = self._direct_plans.get(key)
                if k is None:
                    # (the plan holds raw addresses: the buffers are this op's own -- _buffers -- and live as long as it does)
                    k = self._direct_plans[key] = (a2a, a2a.plan(recv.view(-1), send.view(-1), list(out_splits), list(in_splits)))
                k[0].run(k[1])
                return None
        if self.exchange_mode == "async":
            return dist.all_to_all_single(recv, send, out_splits, in_splits, group=self.group, async_op=True)
        dist.all_to_all_single(recv, send, out_splits, in_splits, group=self.group, async_op=False)
        return None

    def step(self, pl):
        """One whole planned step on the current stream: pool -> all-to-all -> interaction (no overlap between steps)."""
        return self.run_finish(pl, self.run_start(pl))

    def capture_step(self, pl):
        """The planned step as a HIP graph (torch.cuda.CUDAGraph over the library's launches on torch's capture stream):
        a replay costs one host call instead of the Python / ctypes marshalling of two launches and a collective.
        Captured after two eager runs on a side stream (first-use allocations inside the library must not happen
        during capture).  Steps without an exchange only (one rank, or every table replicated): see below."""
        if self.exchange_mode == "p2p" and self.any_sharded:
            raise RuntimeError("capture_step: the p2p exchange counts the uses of a slot in its flag words (a new value per step): not captured")
        if self._exchanges() and self.any_sharded:
            # measured on this stack (ROCm 7.0 / RCCL 2.26, one rank with the exchange forced): capturing the step with its
            # all_to_all_single inside ends in a segmentation fault at replay -- refused here, the bench falls back to the eager loop
            raise RuntimeError("capture_step: a step that holds the RCCL all-to-all is not captured on this stack")
        side = torch.cuda.Stream(device=self.backend.device)
        side.wait_stream(torch.cuda.current_stream(self.backend.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self.step(pl)
        torch.cuda.current_stream(self.backend.device).wait_stream(side)
        torch.cuda.synchronize(self.backend.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.step(pl)
        return g

    # the planned step's pool + exchange on a side stream under the previous batch's interaction (run_start).  Built for the
    # reference's overlap (dlrm_s_pytorch.py:564-569) once the exchange had left the host path (exchange_mode "direct"); measured
    # on one rank with the exchange forced: 149 us per step against 40.5 in stream order, 33.6 against 24.7 without an exchange --
    # the two cross-stream event waits per step cost far more than the 13.7 us exchange kernel they could hide.  Off.
    overlap = False
    _ov = None

    def _overlap_state(self):
        if self._ov is None:
            dev = self.backend.device
            side = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(dev)
            ev = {"side": side, "ready": [torch.cuda.Event() for _ in range(2)], "free": [torch.cuda.Event() for _ in range(2)],
                  "sig": None, "n_ready": [0, 0], "n_free": [0, 0]}
            for e in ev["ready"]:
                e.record(side)     # (torch creates the HIP event at the first record: here, not inside a timed step)
            for e in ev["free"]:
                e.record(cur)
            if self.overlap == "signals":   # two signal words per slot: [ready, free] (8 bytes of signal memory each)
                import ctypes as C_
                sig = []
                try:
                    with torch.cuda.device(dev):
                        for _ in range(2):
                            a, b = C_.c_void_p(), C_.c_void_p()
                            _lib.check(_lib.lib().evs_signal_alloc(C_.byref(a)))
                            _lib.check(_lib.lib().evs_signal_alloc(C_.byref(b)))
                            sig.append((a, b))
                    ev["sig"] = sig
                except Exception:
                    ev["sig"] = None   # (no stream wait-value operations on this device: events)
            self._ov = ev
        return self._ov

    def run_finish(self, pl, work):
        if "route_dst" in pl:
            self._reroute(pl)
        ov_slot = None
        if isinstance(work, tuple) and work and work[0] == "overlap":
            ov_slot, work = work[1], None
            if self._ov["sig"] is not None:
                _lib.check(_lib.lib().evs_stream_wait_value(_stream_ptr(self.backend.device), self._ov["sig"][ov_slot][0], self._ov["n_ready"][ov_slot]))
            else:
                torch.cuda.current_stream(self.backend.device).wait_event(self._ov["ready"][ov_slot])
        if work is not None:
            work.wait()
        tr = self.trace
        if tr is not None and (tr["n"] if self.any_sharded else self._bump(tr)) % tr["every"]:
            tr = None
        p2p = self.exchange_mode == "p2p" and self.any_sharded
        if p2p:
            self._p2p_state(pl["Bg"]).begin_consume(pl["slot"])
        if tr is not None:
            tr["interact"].append(self._stamp())
        R = self.backend.interact_mixed(pl["x"], pl["specs"], self.ev, self.d, self.itself, out=pl["R"], planned=True)
        if tr is not None:
            tr["interact"].append(self._stamp())
        if p2p:
            self._p2p_state(pl["Bg"]).end_consume(pl["slot"])
        if ov_slot is not None:
            if self._ov["sig"] is not None:
                self._ov["n_free"][ov_slot] += 1
                _lib.check(_lib.lib().evs_stream_write_value(_stream_ptr(self.backend.device), self._ov["sig"][ov_slot][1], self._ov["n_free"][ov_slot]))
            else:
                self._ov["free"][ov_slot].record(torch.cuda.current_stream(self.backend.device))
        return R

    # bench: {"pool": [], "interact": [], "n": 0, "every": 4} -> HIP events around the two launches of every 4th step (start,
    # end, start, ...).  Every step would make the loop host-bound (an event record costs about as much as a launch) and
    # an interval then holds the GPU's wait for the host: 211 instead of 13 us for the pool kernel, measured.
    trace = None

    @staticmethod
    def _bump(tr):
        tr["n"] += 1
        return tr["n"]

    @staticmethod
    def _stamp():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def forward(self, x_local, lS_o, lS_i):
        """lS_o/lS_i: per-table offsets/indices of the FULL batch (every rank sees all of them,
        dlrm_s_pytorch.py:543-545); x_local: this rank's (B/world, d) dense features."""
        return self.finish(self.start(lS_o, lS_i), x_local, lS_o, lS_i)


# ----------------------------------------------------------------------------- the extension's own RCCL communicator
_direct = {}


def direct_comm(group, device):
    """The DirectA2A of (group, device) -- made once: rank 0 draws the RCCL unique id, the process group hands it round
    (broadcast_object_list: any backend), every rank joins ncclCommInitRank -- or None when the extension / RCCL is missing on
    ANY rank (agreed over the group first: a communicator some ranks never join would hang the others) or the group is not
    an RCCL one.  EVS_DIRECT_A2A=0 switches it off; EVS_DIRECT_A2A_V=0 takes grouped ncclSend / ncclRecv instead of ncclAllToAllv."""
    dev = torch.device(device)
    key = (id(group) if group is not None else None, dev.index)
    if key in _direct:
        return _direct[key]
    a2a = None
    try:
        from . import _ext
        X = _ext.ext()
        ok = (os.environ.get("EVS_DIRECT_A2A", "1") != "0" and X is not None and hasattr(X, "DirectA2A") and X.rccl_available()
              and dist.is_initialized() and dist.get_backend(group) == "nccl" and dev.type == "cuda")
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            ok = bool(int(t.item()))
        if ok:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            box = [X.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            a2a = X.DirectA2A(box[0], rank, world, dev.index or 0, os.environ.get("EVS_DIRECT_A2A_V", "1") != "0")
    except Exception as ex:   # (never fatal: the collective through torch.distributed is still there)
        import warnings
        warnings.warn("the extension's RCCL communicator could not be made (%r): all_to_all_single is used" % (ex,))
        a2a = None
    _direct[key] = a2a
    return a2a


def direct_close():
    """destroy the extension's communicators (before the process group goes: bench.py, tests)"""
    for a in _direct.values():
        if a is not None:
            a.close()
    _direct.clear()


# ----------------------------------------------------------------------------- bench (N > 1)
def _bench_weights(ln_emb, d, rank, world, dev, owner):
    """the bench's tables of one rank: same values on every rank that holds table t (a row-split table: this rank's row range)"""
    g = torch.Generator(device=dev)
    weights = {}
    for t in [t for t in range(len(ln_emb)) if owner[t] in (rank, -1, -2)]:
        g.manual_seed(1000 + t)
        a = float(np.sqrt(1.0 / ln_emb[t]))
        lo, hi = row_range(ln_emb[t], rank, world) if owner[t] == -2 else (0, ln_emb[t])
        weights[t] = torch.empty((hi - lo, d), dtype=torch.float32, device=dev).uniform_(-a, a, generator=g)
    return weights


def _verify_for_bench(args, ln_emb, rank, world, dev, policy, budget_rows, other="p2p"):
    """verify_p2p_against_collective on the bench's own placement and one of its batches"""
    d, Bg = args.dim, args.batch * world
    owner = plan_placement(ln_emb, world, policy, replicate_budget_rows=budget_rows)
    weights = _bench_weights(ln_emb, d, rank, world, dev, owner)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    off = torch.arange(Bg, device=dev, dtype=torch.int64)
    lS_i = [torch.randint(0, n, (Bg,), device=dev, generator=g, dtype=torch.int64) for n in ln_emb]
    ok = verify_p2p_against_collective(ln_emb, d, rank, world, weights, HipBackend(dev), policy, [off] * len(ln_emb), lS_i,
                                       force_exchange=bool(getattr(args, "force_exchange", False)), one_index_per_bag=True,
                                       replicate_budget_rows=budget_rows, other=other)
    del weights
    torch.cuda.empty_cache()
    return ok


def _bench_policy(args, ln_emb, rank, world, dev, policy, budget_rows, want_roofline):
    """Time args.steps steps of one placement policy (all ranks in lockstep); returns a dict."""
    if getattr(args, "exchange_mode", "inline") == "auto":
        # "auto": the device-to-device exchange only when one batch through both exchanges gave bit-equal receive buffers on
        # every rank; the RCCL collective otherwise
        import copy
        args = copy.copy(args)
        verified = (world > 1 or bool(getattr(args, "force_exchange", False))) and _verify_for_bench(args, ln_emb, rank, world, dev, policy, budget_rows)
        args.exchange_mode = "p2p" if verified else "inline"
        r = _bench_policy(args, ln_emb, rank, world, dev, policy, budget_rows, want_roofline)
        r["exchange_auto"] = {"picked": args.exchange_mode, "p2p_verified": bool(verified)}
        return r
    d = args.dim
    T = len(ln_emb)
    Bl = args.batch
    Bg = Bl * world
    backend = HipBackend(dev)
    owner = plan_placement(ln_emb, world, policy, replicate_budget_rows=budget_rows)
    g = torch.Generator(device=dev)
    weights = _bench_weights(ln_emb, d, rank, world, dev, owner)
    op = ShardedEmbeddingInteract(ln_emb, d, rank, world, weights, backend, policy=policy, one_index_per_bag=True,
                                  replicate_budget_rows=budget_rows)
    op.force_exchange = bool(getattr(args, "force_exchange", False))
    op.exchange_mode = getattr(args, "exchange_mode", "inline")
    op.overlap = getattr(args, "overlap", False) or False
    direct_used = None
    if op.exchange_mode == "direct":   # made here, outside the timed loops (ncclCommInitRank is a rendezvous)
        a2a_ = direct_comm(None, dev) if (world > 1 or op.force_exchange) else None
        direct_used = None if a2a_ is None else ("ncclAllToAllv" if a2a_.use_alltoallv else "grouped ncclSend/ncclRecv")
        # no box with two GPUs has run it: ONE batch through all_to_all_single and through the extension's call, every rank's
        # receive buffer bit for bit, the verdict agreed on over the group -- only then is it the timed exchange
        if a2a_ is not None and not _verify_for_bench(args, ln_emb, rank, world, dev, policy, budget_rows, other="direct"):
            direct_used = "refused: receive buffers differ from all_to_all_single's (%s)" % (getattr(verify_p2p_against_collective, "last_error", None),)
            a2a_ = None
        if a2a_ is None:
            op.exchange_mode = "inline"
    # every rank generates the same full-batch indices (same seed), as the reference feeds them
    g.manual_seed(7)
    nb = 4
    batches = []
    off = torch.arange(Bg, device=dev, dtype=torch.int64)
    for _ in range(nb):
        lS_i = [torch.randint(0, n, (Bg,), device=dev, generator=g, dtype=torch.int64) for n in ln_emb]
        batches.append(([off] * T, lS_i))
    x = torch.rand((Bl, d), device=dev)
    F = T + 1
    P = F * (F - 1) // 2
    outs = [torch.empty((Bl, d + P), device=dev) for _ in range(2)]

    # nb batches x 2 pipeline slots, planned once (serving loops re-use their staged buffers)
    plans = {(j, sl): op.plan(x, batches[j][0], batches[j][1], out=outs[sl], slot=sl) for j in range(nb) for sl in (0, 1)}

    # mode of the timed loop: "pipelined" (default) = eager, the exchange of batch i+1 in flight under the interaction of
    # batch i; "graph" = every planned step captured once as a HIP graph and replayed (--sharded-mode graph; with several
    # ranks the RCCL collective is captured with the kernels).  Measured with one rank (Kaggle, B = 16 384): eager 29.4 us
    # per step, graph replay 33.9 us -- torch's CUDAGraph.replay() costs more host time than the two ctypes launches it
    # replaces, and the kernels inside a graph keep their boundaries.
    mode = getattr(args, "sharded_mode", "auto")
    if mode == "auto":
        mode = "pipelined"
    graphs = None
    if mode == "graph":
        try:
            graphs = [op.capture_step(plans[(j, 0)]) for j in range(nb)]
        except Exception as ex:   # capture not possible here: the eager loop still measures the path
            graphs, mode = None, "pipelined (graph capture failed: %r)" % (ex,)

    def run(steps):
        if graphs is not None:
            for i in range(steps):
                graphs[i % nb].replay()
            return
        # two-deep software pipeline: exchange of batch i+1 overlaps the interaction of batch i
        h = op.run_start(plans[(0, 0)])
        for i in range(steps):
            nxt = None
            if i + 1 < steps:
                nxt = op.run_start(plans[((i + 1) % nb, (i + 1) % 2)])
            op.run_finish(plans[(i % nb, i % 2)], h)
            h = nxt

    if op.exchange_mode == "p2p" and world > 1:
        # first steps over mappings no test has seen (two physical GPUs): all ranks agree that the hand-overs arrive before
        # the long loops start -- a wait that ran out of patience sets the sticky flag on the rank that waited
        run(2)
        op.p2p_flush()
        torch.cuda.synchronize()
        ok = torch.tensor([1 if _lib.lib().evs_check_index_errors(None) == 0 else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise RuntimeError("p2p exchange: a hand-over did not arrive on some rank (%s)" % _lib.lib().evs_last_error())
    run(args.warmup)
    run(1500)   # clock settle: a FIXED number of untimed steps (every rank must issue the same collectives)
    op.p2p_flush()
    e_end = torch.cuda.Event()
    e_end.record()             # (torch creates the HIP event at the first record: tens of microseconds that do not belong in a 20-step region)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    op.p2p_flush()
    e_end.record()
    while not e_end.query():   # (poll, then the closing synchronise: a blocking one wakes up tens of microseconds late)
        pass
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0   # this rank's K steps, from the common start to its own completion; the job's time is the MAX below
    dist.barrier()
    # every rank's own time for the K steps (the job's time is their MAX; the spread shows a placement's imbalance)
    mine = torch.tensor([dt], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    dt_per_rank = [float(t.item()) for t in every]
    dt = max(dt_per_rank)
    n_sh = sum(1 for t in range(T) if owner[t] >= 0)
    n_split = sum(1 for t in range(T) if owner[t] == -2)
    n_rep = T - n_sh - n_split
    # what the placement asks of each rank per step: lookups pooled for the GLOBAL batch (its own sharded tables, its row range
    # of every row-split table) and lookups gathered inside its interaction launch for its LOCAL batch (the replicated tables)
    pool_lookups = [Bg * (len(op.own[r]) + n_split) for r in range(world)]
    local_lookups = [Bl * n_rep for _ in range(world)]
    res = {"dt": dt, "dt_per_rank": dt_per_rank, "pool_lookups_per_rank": pool_lookups, "local_lookups_per_rank": local_lookups,
           "owner": owner, "n_sharded": n_sh, "n_rowsplit": n_split, "n_replicated": n_rep, "mode": mode,
           "exchange_used": op.exchange_mode, "direct_a2a": direct_used, "overlap": bool(op.overlap and op._ov is not None),
           # bytes that leave a rank per step (its pooled vectors -- and its partials of the row-split tables -- for the
           # other ranks' batch slices), and over all ranks
           "a2a_bytes_per_rank": 4 * d * Bl * (len(op.my_own) + n_split) * (world - 1),
           "a2a_bytes": 4 * d * Bl * (n_sh + n_split * world) * (world - 1), "roofline": None}
    # roofline of the launches a rank issues per batch (rank 0, HIP events on the launch stream, in a repeat of the timed
    # loop right behind it): the pooling gather of its own tables over the GLOBAL batch (if any) and the interaction over its
    # LOCAL batch (received pooled vectors dense, replicated tables gathered inside the kernel)
    tr = None
    if want_roofline:
        # the SAME loop as the timed one (pipelined: exchange of batch i+1 in flight under the interaction of batch i, no
        # synchronise between steps; every rank runs it -- it holds the collectives), with HIP events around the two
        # launches of every step; read after the loop
        iters = max(40, min(args.steps, 400))
        torch.cuda.synchronize()
        op.trace = {"pool": [], "interact": [], "n": 0, "every": 4}
        if graphs is None:
            run(iters)
        else:   # (graph mode: the launches are inside the graph; the eager step shows the kernels)
            for i in range(iters):
                op.step(plans[(i % nb, 0)])
        torch.cuda.synchronize()
        tr, op.trace = op.trace, None
    if tr is not None and rank == 0:
        from bench import HBM_PEAK_GBPS as peak
        n_own = len(op.my_own)
        pair_ms = lambda v: sum(v[k].elapsed_time(v[k + 1]) for k in range(0, len(v), 2)) / max(1, len(v) // 2)
        t_pool, t_fin = pair_ms(tr["pool"]), pair_ms(tr["interact"])
        # row + index read, pooled vector written (one index/bag); a row-split table: every index read, 1/world of the rows
        # read, every (mostly zero) partial written
        pool_bytes = n_own * Bg * (4 * d + 8 + 4 * d) + n_split * Bg * (8 + 4 * d // world + 4 * d)
        fin_bytes = Bl * (4 * d * (1 + n_sh) + n_rep * (4 * d + 8) + n_split * (4 * d + 8 + 8) + 4 * (d + P))
        dom = ("pool", pool_bytes, t_pool) if ((n_own or n_split) and t_pool >= t_fin) else ("interact", fin_bytes, t_fin)
        res["roofline"] = {
            "bound": "hbm",
            "kernel": "embedding_bag_sum_kernel (own tables x global batch)" if dom[0] == "pool" else
                      "fused gather + interaction kernel (local batch: %d received dense features + %d replicated tables + %d row-split tables selected from the received partials)" % (n_sh, n_rep, n_split),
            "achieved": dom[1] / dom[2] / 1e6, "peak": peak, "unit": "GB/s", "frac": dom[1] / dom[2] / 1e6 / peak,
            "traffic": None, "bytes_per_launch": dom[1], "avg_launch_ms": dom[2],
            "pool": {"ms": t_pool, "bytes": pool_bytes, "tables": n_own, "row_ranges": n_split}, "interact": {"ms": t_fin, "bytes": fin_bytes}}
    del op, plans, weights, graphs
    torch.cuda.empty_cache()
    return res


def verify_p2p_against_collective(ln_emb, d, rank, world, weights, backend, policy, lS_o, lS_i, force_exchange=False, other="p2p", **op_kw):
    """Before anybody trusts (or times) the device-to-device exchange on hardware no test has seen: ONE batch through both
    exchanges -- the RCCL all_to_all_single and exchange_mode "p2p" -- and every rank's receive buffer compared bit for bit;
    the verdict is agreed on over the process group (all ranks return the same bool, none raises).  Cost: two ops' buffers
    and one step each."""
    ok = 1
    verify_p2p_against_collective.last_error = None   # this rank's reason when the verdict is False because something raised
    try:
        recvs = []
        for mode in ("inline", other):    # (other = "direct", round 6: the extension's own ncclAllToAllv against all_to_all_single)
            op = ShardedEmbeddingInteract(ln_emb, d, rank, world, weights, backend, policy=policy, **op_kw)
            op.force_exchange = bool(force_exchange)
            op.exchange_mode = mode
            if not op.any_sharded:
                del op
                return True   # nothing is exchanged under this placement
            work, recv, Bg, Bl, out_splits = op.start(lS_o, lS_i)
            if work is not None:
                work.wait()
            st_, slot = op._p2p_slot_of(recv, Bg) if mode == "p2p" else (None, None)
            if slot is not None:
                st_.begin_consume(slot)      # every source's block has arrived
                st_.end_consume(slot)
                op.p2p_flush()
            torch.cuda.synchronize(backend.device)
            n = sum(out_splits)
            recvs.append(recv.view(-1)[:n].clone())
            if mode == "p2p":
                if _lib.lib().evs_check_index_errors(None) != 0:
                    ok = 0               # a hand-over that did not arrive
                for st2 in op._p2p.values():
                    st2.close()
            del op
        if ok and not torch.equal(recvs[0], recvs[1]):
            ok = 0
    except Exception as ex:
        ok = 0
        verify_p2p_against_collective.last_error = repr(ex)
    t = torch.tensor([ok], device=backend.device, dtype=torch.int32)
    try:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    except Exception:
        return False
    return bool(int(t.item()))


def bench_p2p_side(args, ln_emb, rank, world, dev):
    """The same placement and steps with exchange_mode "p2p" (no collective call in the step), as a side line of the N > 1
    bench: never raises, and every rank returns from it (set-up failures and hand-overs that do not arrive are agreed on over
    the process group before anybody proceeds)."""
    import copy
    try:
        a = copy.copy(args)
        a.exchange_mode = "p2p"
        policy = getattr(args, "placement", "rows+replicate")
        budget_rows = int(getattr(args, "replicate_gb", 64.0) * 1e9 / (4 * args.dim))
        verified = _verify_for_bench(args, ln_emb, rank, world, dev, policy, budget_rows)
        if not verified:
            why = getattr(verify_p2p_against_collective, "last_error", None)
            if why:   # the exchange could not even be set up / run on this rank (or a peer said so in the hand-shake)
                return {"verified": False, "error": why}
            return {"verified": False, "skipped": "one batch through both exchanges: the p2p receive buffers differ from the RCCL ones (or a "
                                                  "hand-over did not arrive) on some rank -- not timed; the RCCL line above stands"}
        e = _bench_policy(a, ln_emb, rank, world, dev, policy, budget_rows, True)
        if e["n_sharded"] == 0 and not e["n_rowsplit"]:
            return {"skipped": "nothing is exchanged under this placement"}
        T, Bg = len(ln_emb), args.batch * world
        return {"verified": True, "value": T * Bg * args.steps / e["dt"], "unit": "lookups/s", "ms_per_step": e["dt"] / args.steps * 1e3, "roofline": e["roofline"],
                "note": "the same step without a collective call: the pooling launch writes every peer's block into that peer's IPC-mapped "
                        "receive buffer over xGMI, two flag words per (peer, slot) hand it over (csrc/evs_p2p.hip)"}
    except Exception as ex:
        return {"error": repr(ex)}


def bench_sharded(args, ln_emb, rank, world, dev):
    """Weak scaling: global batch = world * args.batch; every rank times the same K steps.
    The HEADLINE placement is BASELINE.json's: tables sharded by row count across the ranks with ONE RCCL
    all_to_all_single of pooled vectors per batch ("rows+replicate": tables of at most 1 M rows are replicated and looked
    up locally, so their pooled vectors never cross xGMI; "rows" shards every table).  The memory-aware placement
    ("hbm": replicate whatever fits --replicate-gb per GPU; for the 4.9 GB Kaggle model that is everything, i.e. pure
    data parallel with no exchange) is timed beside it in the same run as `replicated_all`."""
    d = args.dim
    T = len(ln_emb)
    Bl = args.batch
    Bg = Bl * world
    policy = getattr(args, "placement", "rows+replicate")
    budget_rows = int(getattr(args, "replicate_gb", 64.0) * 1e9 / (4 * d))
    main = _bench_policy(args, ln_emb, rank, world, dev, policy, budget_rows, True)
    extra = None
    if policy != "hbm":   # the no-exchange alternative, when the model fits the per-GPU budget
        try:
            if sum(ln_emb) <= budget_rows:
                e = _bench_policy(args, ln_emb, rank, world, dev, "hbm", budget_rows, True)
                extra = {"value": T * Bg * args.steps / e["dt"], "unit": "lookups/s", "ms_per_step": e["dt"] / args.steps * 1e3,
                         "placement": "hbm (every table replicated on every GPU: %.1f GB of 288 GB; data parallel, no exchange step)"
                                      % (sum(ln_emb) * 4 * d / 1e9),
                         "mode": e["mode"], "roofline": e["roofline"]}
            else:
                extra = {"skipped": "the model (%.1f GB) exceeds the replication budget of %.0f GB per GPU"
                                    % (sum(ln_emb) * 4 * d / 1e9, getattr(args, "replicate_gb", 64.0))}
        except Exception as ex:  # the headline number must survive a failure of the side measurement
            extra = {"error": repr(ex)}
    dt = main["dt"]
    lookups = T * Bg
    per_rank = {"step_ms": [t / args.steps * 1e3 for t in main["dt_per_rank"]],
                "pool_lookups_per_step": main["pool_lookups_per_rank"],           # own sharded tables (+ row ranges) x the GLOBAL batch
                "local_lookups_per_step": main["local_lookups_per_rank"],         # replicated tables x the LOCAL batch, inside the interaction launch
                "tables_owned": [sum(1 for o in main["owner"] if o == r) for r in range(world)],
                "note": "a rank's step = pool launch over its own tables for the global batch (nothing when it owns none), the exchange, "
                        "the interaction over its local batch; value uses the MAX of step_ms"}
    pl_ = per_rank["pool_lookups_per_step"]
    per_rank["pool_imbalance_max_over_mean"] = (max(pl_) / (sum(pl_) / len(pl_))) if sum(pl_) else None
    single = None
    if world == 1:
        # the same launcher on ONE rank beside the plain single-process step (one fused launch, every table local): the sharded
        # code path must stay within 25 % of it
        try:
            single = _single_process_line(args, ln_emb, dev)
            single["sharded_over_single"] = (lookups * args.steps / dt) / single["value"]
        except Exception as ex:
            single = {"error": repr(ex)}
    shape = "Criteo-Kaggle" if sum(ln_emb) < 100_000_000 else "Criteo-Terabyte-shaped"
    if main.get("n_rowsplit"):
        what = ("%d tables split ROW-WISE over the %d ranks (each rank pools its row range for the global batch; the receiver "
                "selects the partial of the rank that holds the row) + %d replicated + %d whole tables sharded (%s), one "
                "all_to_all_single per batch over RCCL/xGMI" % (main["n_rowsplit"], world, main["n_replicated"], main["n_sharded"], policy))
        par = "row-split x%d + a2a" % world
    elif main["n_sharded"] == 0:
        what = ("all %d tables replicated on every GPU (%.1f GB of 288 GB): pure data parallel, no exchange step"
                % (T, sum(ln_emb) * 4 * d / 1e9))
        par = "replicated tables x%d (data parallel)" % world
    else:
        what = ("%d tables sharded by rows + %d replicated (%s), one all_to_all_single of pooled vectors per batch over "
                "RCCL/xGMI" % (main["n_sharded"], main["n_replicated"], policy))
        par = "table-sharded x%d + a2a" % world
    return {
        "metric": "inference lookups/sec, %s 26-table DLRM (apply_emb + interact_features)" % shape,
        "value": lookups * args.steps / dt, "unit": "lookups/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s 26 tables (%.1f M rows) x d=%d fp32 over %d GPUs: %s; uniform indices, one index per bag"
                               % (shape, sum(ln_emb) / 1e6, d, world, what),
                   "batch_per_gpu": Bl, "global_batch": Bg, "tables": T, "dim": d, "parallelism": par,
                   "placement": policy, "owner": main["owner"], "step_mode": main["mode"],
                   "observed_world_size": dist.get_world_size(), "backend": dist.get_backend(),
                   "exchange_mode": (main.get("exchange_auto") or {}).get("picked", main.get("exchange_used", getattr(args, "exchange_mode", "inline"))),
                   "exchange_requested": getattr(args, "exchange_mode", "inline"), "direct_a2a": main.get("direct_a2a"),
                   "overlap": main.get("overlap"),
                   "exchange_auto": main.get("exchange_auto"),
                   "a2a_bytes_per_step_per_rank": main["a2a_bytes_per_rank"],
                   "a2a_bytes_per_step_all_links": main["a2a_bytes"]},
        "roofline": main["roofline"], "cpu_baseline": None, "replicated_all": extra, "per_rank": per_rank,
        "single_process": single,
    }


def _single_process_line(args, ln_emb, dev):
    """the plain N = 1 step timed in this process, for the line a one-rank sharded run prints beside its own: `value` = bench.py's
    N = 1 headline form (ONE fused launch over local tables, lS_o GIVEN and checked in the kernel: what `python3 bench.py --gpus 1`
    reports and what a scaling efficiency is computed against), `declared_one_index` = the same launch with one index per bag
    declared, as the sharded step declares it"""
    from . import dlrm_ops
    d, B, T = args.dim, args.batch, len(ln_emb)
    owner = [0] * T
    w = _bench_weights(ln_emb, d, 0, 1, dev, owner)
    ev = dlrm_ops.EVTables.from_fp32([w[t] for t in range(T)], device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    off = torch.arange(B, device=dev, dtype=torch.int64).repeat(T, 1)
    bs = [torch.stack([torch.randint(0, n, (B,), device=dev, generator=g, dtype=torch.int64) for n in ln_emb]) for _ in range(4)]
    x = torch.rand((B, d), device=dev)
    F = T + 1
    out = [torch.empty((B, d + F * (F - 1) // 2), device=dev) for _ in range(2)]

    def timed(declared):
        def run(n):
            for i in range(n):
                dlrm_ops.apply_emb_interact(x, off, bs[i % 4], ev, None, out=out[i % 2], one_index_per_bag=declared)

        run(args.warmup + 1500)
        e = torch.cuda.Event()
        e.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        e.record()
        while not e.query():
            pass
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dt, dt_decl = timed(False), timed(True)
    del ev, w
    torch.cuda.empty_cache()
    return {"value": T * B * args.steps / dt, "unit": "lookups/s", "ms_per_step": dt / args.steps * 1e3,
            "declared_one_index": {"value": T * B * args.steps / dt_decl, "ms_per_step": dt_decl / args.steps * 1e3},
            "note": "one fused launch per step in this process (no placement, no exchange): what `bench.py --gpus 1` times (lS_o given); "
                    "a sharded step is two launches -- the pooling gather of the rank's own tables into the exchange layout, then the "
                    "interaction over received + replicated features"}
