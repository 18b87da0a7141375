"""GPU-resident cache tier (EvLFU / LRU / LFU) -- Python handle over the C ABI (evs_cache_*).

Reference policies: cache_algo/EvLFU_C1.py, LRU.py, LFU.py; C++/Cython EvLFU variants differ
only in three constants (see include/evstore_hip.h).  All state (hash table, priority lists,
row arena) lives in HBM; this class only owns device buffers and marshals pointers.
"""
import ctypes as C

import torch

from . import _ext, _lib

POLICY = {"evlfu": 0, "lru": 1, "lfu": 2}
# (flush_rate, perfect_item_cap, flush_extra, perfect_mode)
EVLFU_VARIANTS = {"python": (0.3, 0.95, 1, 0), "cpp": (0.3, 0.95, 0, 2), "cython": (0.4, 1.0, 1, 1)}


def _dev_ptr(t):
    """Device-side address of a device tensor or of a pinned host tensor."""
    if t.is_cuda:
        return t.data_ptr()
    assert t.is_pinned() and t.is_contiguous(), "host tensors must be pinned (torch.Tensor.pin_memory) and contiguous"
    p = _lib.lib().evs_host_device_pointer(t.data_ptr())
    if not p:
        raise _lib.EvsError(_lib.EVS_EINVAL, "pinned tensor is not device-accessible")
    return p


def _check_batch(cache, rows, out=None, hit=None, out_cols=None, pinned_ok=False):
    """Shapes the kernels rely on (a wrong one is an out-of-bounds device access, not an exception): rows (B, n_tables)
    int32 contiguous; out fp32 contiguous with B * out_cols elements (default n_tables * dim); hit / tier uint8 contiguous
    with B * n_tables elements.  Device tensors -- or, for the exact path, pinned host tensors.  -> B"""
    def where(t):
        return t.is_cuda or (pinned_ok and t.is_pinned())
    if not (rows.dtype == torch.int32 and rows.dim() == 2 and rows.shape[1] == cache.n_tables and rows.is_contiguous() and where(rows)):
        raise ValueError("rows must be a contiguous (B, %d) int32 %s tensor, got %s %s" %
                         (cache.n_tables, "device or pinned host" if pinned_ok else "device", tuple(rows.shape), rows.dtype))
    B = int(rows.shape[0])
    if out is not None:
        n = B * (out_cols if out_cols is not None else cache.n_tables * cache.dim)
        if not (out.dtype == torch.float32 and out.is_contiguous() and out.numel() == n and where(out)):
            raise ValueError("out must be a contiguous fp32 tensor of %d elements, got %s %s" % (n, tuple(out.shape), out.dtype))
    if hit is not None:
        if not (hit.dtype == torch.uint8 and hit.is_contiguous() and hit.numel() == B * cache.n_tables and where(hit)):
            raise ValueError("hit / tier must be a contiguous uint8 tensor of %d elements, got %s %s" %
                             (B * cache.n_tables, tuple(hit.shape), hit.dtype))
    return B


class FileTier:
    """File-backed miss tier (evs_filetier_*): ev-table-N.bin files mapped read-only; tables registered with the GPU
    smallest first while they fit pinned_budget_bytes (read zero-copy by the kernels), the rest served through the
    host's reader pool (mmap_file_read.py:32-40 semantics: row r at byte row_bytes * r)."""

    def __init__(self, paths, row_bytes, pinned_budget_bytes):
        self.paths, self.row_bytes, self.n_tables = list(paths), int(row_bytes), len(paths)
        arr = (C.c_char_p * self.n_tables)(*[p.encode() for p in self.paths])
        h = C.c_void_p()
        _lib.check(_lib.lib().evs_filetier_open(C.byref(h), self.n_tables, arr, self.row_bytes, int(pinned_budget_bytes)))
        self._h = h
        rows = (C.c_int64 * self.n_tables)()
        ptrs = (C.c_void_p * self.n_tables)()
        reg = (C.c_int * self.n_tables)()
        pinned = C.c_int64()
        _lib.check(_lib.lib().evs_filetier_info(self._h, rows, ptrs, reg, C.byref(pinned)))
        self.n_rows = [int(v) for v in rows]
        self.registered = [bool(v) for v in reg]
        self.pinned_bytes = int(pinned.value)

    def fetch(self, keys):
        """the reader pool: keys (n,) uint64 numpy array of (table_1based << 32 | row) -> (n, row_bytes) uint8"""
        import numpy as np
        keys = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros((len(keys), self.row_bytes), np.uint8)
        _lib.check(_lib.lib().evs_filetier_fetch(self._h, len(keys), keys.ctypes.data, out.ctypes.data, 0))
        return out

    def close(self):
        if self._h:
            _lib.lib().evs_filetier_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_hwq_warned = False


def _warn_hw_queues():
    """the resident server wants a hardware queue nothing else is folded onto (evstore_dlrm_amd.configure_runtime): say so ONCE
    when the process runs on the runtime's default of four -- the package no longer sets the knob at import"""
    global _hwq_warned
    if _hwq_warned:
        return
    _hwq_warned = True
    import os
    try:
        n = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        n = 4
    if n < 8:
        import warnings
        warnings.warn("GpuCache.serve_start: GPU_MAX_HW_QUEUES=%d -- a copy or kernel of the caller's that HIP folds onto the resident "
                      "server's hardware queue waits until the server goes home idle (up to idle_us per request); call "
                      "evstore_dlrm_amd.configure_runtime() before the first GPU call, or export GPU_MAX_HW_QUEUES=8" % n)


class GpuCache:
    def __init__(self, policy, capacity, n_tables=26, dim=36, codec=32, variant="python", device="cuda"):
        self.policy, self.capacity, self.n_tables, self.dim, self.codec = policy, int(capacity), n_tables, dim, codec
        self.device = torch.device(device)
        fr, pc, ex, pm = EVLFU_VARIANTS[variant]
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().evs_cache_create(C.byref(h), POLICY[policy], self.capacity, n_tables, dim, codec,
                                                   fr, pc, ex, pm))
        self._h = h
        self._backing = None

    def __del__(self):
        try:
            if self._h:
                _lib.lib().evs_cache_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _dev_index(self):
        return torch.cuda.current_device() if self.device.index is None else self.device.index

    def set_backing(self, tables):
        """tables: EVTables, or a list of uint8/float tensors (HBM or pinned host memory) in the cache's codec."""
        raws = tables.raw if hasattr(tables, "raw") else list(tables)
        assert len(raws) == self.n_tables
        rb = self.dim * self.codec // 8
        n_rows = [int(t.numel() * t.element_size() // rb) for t in raws]
        self._backing = raws  # keep alive
        ptrs = (C.c_void_p * self.n_tables)(*[_dev_ptr(t) for t in raws])   # pinned host tables: their device-side address
        rows = (C.c_int64 * self.n_tables)(*n_rows)
        _lib.check(_lib.lib().evs_cache_set_backing(self._h, ptrs, rows))

    def set_file_backing(self, tier):
        """tier: FileTier -- the miss tier is a set of .bin files (registered tables zero-copy, the rest staged by the
        host's reader pool; batched lookups only when any table is staged)."""
        assert tier.n_tables == self.n_tables and tier.row_bytes == self.dim * self.codec // 8
        self._backing = tier  # keep alive
        _lib.check(_lib.lib().evs_cache_set_file_backing(self._h, tier._h))

    def staged_rows(self):
        return int(_lib.lib().evs_cache_staged_rows(self._h))

    def set_batch_policy(self, policy):
        """'sampled' (one update kernel, victim = lowest priority of 8 sampled entries), 'plan' (insert / plan / evict /
        assign, clock-hand window) or 'setassoc' (8-way set-associative: one 64-byte line of key words per set, the
        victim is the lowest priority of the key's own set; single tier, tables in HBM); before the first batched lookup."""
        _lib.check(_lib.lib().evs_cache_set_batch_policy(self._h, {"plan": 0, "sampled": 1, "setassoc": 2}[policy]))
        return self

    def request(self, rows, approx_thres=-1, out=None, hit=None):
        """rows: (B, n_tables) int32 tensor.  Requests are replayed strictly in order.
        Returns (hit (B,T) uint8, out (B,T,dim) fp32).
        rows / out / hit may be device tensors or PINNED host tensors (torch pin_memory): pinned buffers are
        read and written by the kernel itself, so the reference's one-request-at-a-time loop costs one launch
        and one synchronise per request instead of two copies around it (synchronise before reading them)."""
        B = _check_batch(self, rows, out, hit, pinned_ok=True)
        if out is None:
            out = torch.empty((B, self.n_tables, self.dim), dtype=torch.float32, device=self.device)
        if hit is None:
            hit = torch.empty((B, self.n_tables), dtype=torch.uint8, device=self.device)
        X = _ext.ext()
        if X is not None:
            X.cache_request(self._h.value, rows, out, hit, int(approx_thres), self._dev_index())
            return hit, out
        _lib.check(_lib.lib().evs_cache_request(self._h, B, _dev_ptr(rows), _dev_ptr(out), _dev_ptr(hit),
                                                int(approx_thres), torch.cuda.current_stream(self.device).cuda_stream))
        return hit, out

    # ---- the exact policy as a resident server (include/evstore_hip.h: evs_cache_serve_*) ----
    def serve_start(self, approx_thres=-1, n_slots=4, idle_us=200):
        """Arm the mailbox server: batch-1 requests then cost two cache-line hand-overs over the bus instead of a launch and a
        synchronise each.  The rows of request i land in self.serve_ring[slot] on the DEVICE."""
        import ctypes as C
        import numpy as np
        _warn_hw_queues()
        self.serve_ring = torch.empty((n_slots, self.n_tables, self.dim), dtype=torch.float32, device=self.device)
        self._srv_views = [self.serve_ring[k] for k in range(n_slots)]   # (a tensor index per request is 2 us of a 15 us request)
        self._srv_rows = np.zeros(self.n_tables, np.int32)
        self._srv_hit = np.zeros(self.n_tables, np.uint8)
        self._srv_slot = C.c_int(0)
        self._srv_call = (_lib.lib().evs_cache_serve_request, self._h, self._srv_rows.ctypes.data, self._srv_hit.ctypes.data, C.byref(self._srv_slot))
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().evs_cache_serve_start(self._h, int(approx_thres), self.serve_ring.data_ptr(), int(n_slots), int(idle_us)))
        return self

    def serve_request(self, row_ids):
        """row_ids: n_tables ints (host).  -> (hit flags: a numpy uint8 view valid until the next request, the (T, dim) fp32
        rows as a DEVICE tensor view of a ring slot).  Same results as request() one at a time.
        The server overwrites a slot when the request n_slots later is POSTED (host order): a caller that only ENQUEUES its
        reads of the view (a clone, a kernel) calls serve_consumed() behind them -- the slot is then handed out again only
        after they have run; otherwise the rows must have been read before n_slots - 1 more requests are posted."""
        self._srv_rows[:] = row_ids
        fn, h, rp, hp, sp = self._srv_call
        rc = fn(h, rp, hp, sp)
        if rc:
            _lib.check(rc)
        return self._srv_hit, self._srv_views[self._srv_slot.value]

    def serve_request_to(self, row_ids, out):
        """the same request with the rows written by the server into `out` (a contiguous (T, dim) fp32 DEVICE tensor of the
        caller's, not in use by pending work) instead of a ring slot; row_ids: n_tables ints on the host, or a (T, ...) int64
        device tensor whose element 0 of each row is the id (the reference's lS_i on the GPU).  -> hit flags (numpy uint8 view)"""
        if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == self.n_tables * self.dim):
            raise ValueError("out must be a contiguous fp32 device tensor of %d elements" % (self.n_tables * self.dim))
        L = _lib.lib()
        if torch.is_tensor(row_ids) and row_ids.is_cuda:
            if row_ids.dtype != torch.int64 or row_ids.shape[0] != self.n_tables:
                raise ValueError("device ids: a (T, ...) int64 tensor")
            _lib.check(L.evs_cache_serve_request_to(self._h, None, row_ids.data_ptr(), int(row_ids.stride(0)), out.data_ptr(), self._srv_hit.ctypes.data))
        else:
            self._srv_rows[:] = row_ids
            _lib.check(L.evs_cache_serve_request_to(self._h, self._srv_rows.ctypes.data, None, 0, out.data_ptr(), self._srv_hit.ctypes.data))
        return self._srv_hit

    def serve_consumed(self, slot=None, stream=None):
        """the reads of ring slot `slot` (default: the last request's) have been enqueued on `stream` (default: the current one)"""
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        _lib.check(_lib.lib().evs_cache_serve_consumed(self._h, int(self._srv_slot.value if slot is None else slot), st.cuda_stream))

    def set_inline_update(self, on=True):
        """the set-associative tier's policy update inside the probe + interaction launch (the default where it applies; hit
        flags then mean "served from the cache") or, on=False, the two-launch chain with strict snapshot flags
        (include/evstore_hip.h: evs_cache_set_inline_update)"""
        _lib.check(_lib.lib().evs_cache_set_inline_update(self._h, 1 if on else 0))
        return self

    def serve_stop(self):
        _lib.check(_lib.lib().evs_cache_serve_stop(self._h))

    def lookup_batch(self, rows, out=None, hit=None):
        """Batched EvLFU lookup, snapshot semantics (see include/evstore_hip.h: evs_cache_lookup_batch)."""
        B = _check_batch(self, rows, out, hit)
        if out is None:
            out = torch.empty((B, self.n_tables, self.dim), dtype=torch.float32, device=self.device)
        if hit is None:
            hit = torch.empty((B, self.n_tables), dtype=torch.uint8, device=self.device)
        _lib.check(_lib.lib().evs_cache_lookup_batch(self._h, B, rows.data_ptr(), out.data_ptr(), hit.data_ptr(),
                                                     torch.cuda.current_stream(self.device).cuda_stream))
        return hit, out

    def lookup_interact(self, rows, x, itself=False, out=None, hit=None):
        """R = interact_features(x, cached rows of the B requests): probe + fused MFMA kernel reading the
        rows through a pointer table (no (B,T,d) intermediate), then the batched policy update."""
        F = self.n_tables + 1
        P = F * (F + 1) // 2 if itself else F * (F - 1) // 2
        B = _check_batch(self, rows, out, hit, out_cols=self.dim + P)
        if out is None:
            out = torch.empty((B, self.dim + P), dtype=torch.float32, device=self.device)
        if hit is None:
            hit = torch.empty((B, self.n_tables), dtype=torch.uint8, device=self.device)
        if not (x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (B, self.dim) and x.stride(1) == 1):
            raise ValueError("x must be a (B, %d) fp32 device tensor with unit inner stride" % self.dim)
        X = _ext.ext()
        if X is not None:
            X.cache_lookup_interact(self._h.value, rows, x, bool(itself), out, hit)
            return hit, out
        _lib.check(_lib.lib().evs_cache_lookup_interact(
            self._h, B, rows.data_ptr(), x.data_ptr(), int(x.stride(0)) if B > 1 else self.dim, int(bool(itself)),
            out.data_ptr(), hit.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        return hit, out

    def batch_stats(self):
        s = (C.c_int64 * 8)()
        hist = (C.c_int64 * (self.n_tables + 1))()
        _lib.check(_lib.lib().evs_cache_batch_stats(self._h, s, hist, torch.cuda.current_stream(self.device).cuda_stream))
        keys = ("size", "n_free", "n_tomb", "n_flush", "n_evict", "n_requests", "n_perfect_hits", "n_hits")
        d = dict(zip(keys, [int(v) for v in s]))
        d["hist"] = [int(v) for v in hist]
        return d

    def batch_dump(self):
        import numpy as np
        st = torch.cuda.current_stream(self.device).cuda_stream
        n = _lib.lib().evs_cache_batch_dump(self._h, None, 0, st)
        if n < 0:
            _lib.check(int(n))
        out = np.zeros((max(n, 1), 3), np.int64)
        _lib.lib().evs_cache_batch_dump(self._h, out.ctypes.data, n, st)
        return out[:n]

    def stats(self):
        s = (C.c_int64 * 8)()
        _lib.check(_lib.lib().evs_cache_stats(self._h, s, torch.cuda.current_stream(self.device).cuda_stream))
        keys = ("min_c1", "n_perfect", "size", "n_flush", "n_evict", "n_requests", "n_perfect_hits", "n_hits")
        return dict(zip(keys, [int(v) for v in s]))

    def reset_counters(self):
        _lib.check(_lib.lib().evs_cache_reset_counters(self._h, torch.cuda.current_stream(self.device).cuda_stream))

    def dump(self):
        """Resident keys in list order: rows of (bucket | frequency | 0, table_1based, row)."""
        import numpy as np
        st = torch.cuda.current_stream(self.device).cuda_stream
        n = _lib.lib().evs_cache_dump(self._h, None, 0, st)
        if n < 0:
            _lib.check(int(n))
        out = np.zeros((max(n, 1), 3), np.int64)
        _lib.lib().evs_cache_dump(self._h, out.ctypes.data, n, st)
        return out[:n]


def request_c1c2(c1, c2, rows, threshold=23, out=None, tier=None):
    """Two-tier request (mixed_precs_caching/evlfu_8.cpp:669-796): c1 = main-precision GpuCache, c2 =
    secondary-precision GpuCache (both variant="cpp").  Returns (tier (B,T) uint8, out (B,T,dim) fp32)."""
    B = _check_batch(c1, rows, out, tier)
    if out is None:
        out = torch.empty((B, c1.n_tables, c1.dim), dtype=torch.float32, device=c1.device)
    if tier is None:
        tier = torch.empty((B, c1.n_tables), dtype=torch.uint8, device=c1.device)
    _lib.check(_lib.lib().evs_cache_request_c1c2(c1._h, c2._h, B, rows.data_ptr(), out.data_ptr(), tier.data_ptr(),
                                                 int(threshold), torch.cuda.current_stream(c1.device).cuda_stream))
    return tier, out


def lookup_batch_c1c2(c1, c2, rows, threshold=23, out=None, tier=None, c3=None):
    """Batched two-tier lookup with snapshot semantics (include/evstore_hip.h: evs_cache_lookup_batch_c1c2): the
    throughput form of request_c1c2.  Returns (tier (B,T) uint8: 1 = C1 hit, 2 = C2 hit, 0 = miss; out (B,T,dim) fp32).
    c3 (GpuAltKeyTier): the three-tier form (evs_cache_lookup_batch_c1c2c3) -- tier code 3 = the alt row was served."""
    B = _check_batch(c1, rows, out, tier)
    if out is None:
        out = torch.empty((B, c1.n_tables, c1.dim), dtype=torch.float32, device=c1.device)
    if tier is None:
        tier = torch.empty((B, c1.n_tables), dtype=torch.uint8, device=c1.device)
    st = torch.cuda.current_stream(c1.device).cuda_stream
    if c3 is None:
        _lib.check(_lib.lib().evs_cache_lookup_batch_c1c2(c1._h, c2._h, B, rows.data_ptr(), out.data_ptr(), tier.data_ptr(),
                                                          int(threshold), st))
    else:
        _lib.check(_lib.lib().evs_cache_lookup_batch_c1c2c3(c1._h, c2._h, c3._h, B, rows.data_ptr(), out.data_ptr(),
                                                            tier.data_ptr(), int(threshold), st))
    return tier, out


def lookup_batch_c1c2c3(c1, c2, c3, rows, threshold=23, out=None, tier=None):
    return lookup_batch_c1c2(c1, c2, rows, threshold, out, tier, c3=c3)


def lookup_interact_c1c2c3(c1, c2, c3, rows, x, threshold=23, itself=False, out=None, tier=None, fused=True):
    return lookup_interact_c1c2(c1, c2, rows, x, threshold, itself, out, tier, fused, c3=c3)


def lookup_interact_c1c2(c1, c2, rows, x, threshold=23, itself=False, out=None, tier=None, fused=True, c3=None):
    """The two-tier snapshot lookup with interact_features as its consumer -> (tier, R).  fused (default): every row is
    decoded from the precision of the tier that serves it inside the interaction kernel (evs_cache_lookup_interact_c1c2);
    fused=False: lookup_batch_c1c2 into fp32 (B,T,dim) rows (`out`), then the dense interaction over them."""
    if not fused or c1.dim not in (16, 32, 36):
        from .dlrm_ops import interact_features
        tier, rows_fp32 = lookup_batch_c1c2(c1, c2, rows, threshold, out, tier, c3=c3)
        return tier, interact_features(x, list(rows_fp32.unbind(1)), "dot", itself)
    B = _check_batch(c1, rows, None, tier)
    F = c1.n_tables + 1
    P = F * (F + 1) // 2 if itself else F * (F - 1) // 2
    R = torch.empty((B, c1.dim + P), dtype=torch.float32, device=c1.device)
    if tier is None:
        tier = torch.empty((B, c1.n_tables), dtype=torch.uint8, device=c1.device)
    if not (x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (B, c1.dim) and x.stride(1) == 1):
        raise ValueError("x must be a (B, %d) fp32 device tensor with unit inner stride" % c1.dim)
    xs = int(x.stride(0)) if B > 1 else c1.dim
    st = torch.cuda.current_stream(c1.device).cuda_stream
    if c3 is None:
        _lib.check(_lib.lib().evs_cache_lookup_interact_c1c2(
            c1._h, c2._h, B, rows.data_ptr(), x.data_ptr(), xs, int(bool(itself)), R.data_ptr(), tier.data_ptr(), int(threshold), st))
    else:
        _lib.check(_lib.lib().evs_cache_lookup_interact_c1c2c3(
            c1._h, c2._h, c3._h, B, rows.data_ptr(), x.data_ptr(), xs, int(bool(itself)), R.data_ptr(), tier.data_ptr(),
            int(threshold), st))
    return tier, R


class GpuAltKeyTier:
    """C3: key -> alt-key map with second-chance FIFO (deterministic re-specification of
    mixed_precs_caching/aprx_embedding.cpp).  alt_tables: per table a device uint32 tensor (viewed as int32 is fine)
    with alt_key[row] = alt_row*100 + alt_table_1based."""

    def __init__(self, capacity, alt_tables, device="cuda"):
        self.device = torch.device(device)
        self.n_tables = len(alt_tables)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().evs_aprx_create(C.byref(h), int(capacity), self.n_tables))
        self._h = h
        self._alt = [t.contiguous() for t in alt_tables]
        ptrs = (C.c_void_p * self.n_tables)(*[t.data_ptr() for t in self._alt])
        rows = (C.c_int64 * self.n_tables)(*[int(t.numel()) for t in self._alt])
        _lib.check(_lib.lib().evs_aprx_set_altkeys(self._h, ptrs, rows))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().evs_aprx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def stats(self):
        s = (C.c_int64 * 4)()
        _lib.check(_lib.lib().evs_aprx_stats(self._h, s, torch.cuda.current_stream(self.device).cuda_stream))
        return dict(size=int(s[0]), n_hit=int(s[1]), n_pending=int(s[2]), error=int(s[3]))

    def batch_dump(self):
        """Batched form: (members as an (n,3) int64 array of (table_1based, row, recency flag), stats dict)."""
        import numpy as np
        st = torch.cuda.current_stream(self.device).cuda_stream
        o4 = (C.c_int64 * 4)()
        n = _lib.lib().evs_aprx_batch_dump(self._h, None, 0, o4, st)
        if n < 0:
            _lib.check(int(n))
        out = np.zeros((max(n, 1), 3), np.int64)
        _lib.lib().evs_aprx_batch_dump(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), n, o4, st)
        return out[:n], dict(members=int(o4[0]), n_hit=int(o4[1]), capacity=int(o4[2]))

    def apply_ops(self, ops):
        """APRX_EV's single-key methods in order: ops (n,3) int32 device tensor of (op, table_1based, row) with op 0
        insert_altkey | 1 get_altkey | 2 set_recency_flag | 3 evict_one_key (aprx_embedding.cpp:278-288,341-350,390-411).
        -> uint32-valued int64 tensor: the alt key for op 0 / 1 (0xffffffff = miss), else 0."""
        assert ops.dtype == torch.int32 and ops.is_cuda and ops.is_contiguous() and ops.dim() == 2 and ops.shape[1] == 3
        res = torch.zeros((ops.shape[0],), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().evs_aprx_apply_ops(self._h, int(ops.shape[0]), ops.data_ptr(), res.data_ptr(),
                                                 torch.cuda.current_stream(self.device).cuda_stream))
        return res.to(torch.int64) & 0xffffffff

    def queue(self):
        """the FIFO front to back as (n,2) int64 (table_1based, row), stale duplicates included"""
        st = torch.cuda.current_stream(self.device).cuda_stream
        n = int(_lib.lib().evs_aprx_dump_queue(self._h, None, 0, st))
        if n < 0:
            _lib.check(n)
        out = (C.c_int64 * (2 * max(n, 1)))()
        _lib.lib().evs_aprx_dump_queue(self._h, out, n, st)
        return torch.tensor(list(out)[:2 * n], dtype=torch.int64).view(n, 2)


def request_c1c2c3(c1, c2, c3, rows, threshold=23, out=None, tier=None):
    """request_to_c1_c2_c3 (evlfu_8.cpp:492-667): tier codes 1 C1 hit, 2 C2 hit, 3 alt-key hit, 0 miss."""
    assert rows.dtype == torch.int32 and rows.is_cuda and rows.is_contiguous()
    B = int(rows.shape[0])
    if out is None:
        out = torch.empty((B, c1.n_tables, c1.dim), dtype=torch.float32, device=c1.device)
    if tier is None:
        tier = torch.empty((B, c1.n_tables), dtype=torch.uint8, device=c1.device)
    _lib.check(_lib.lib().evs_cache_request_c1c2c3(c1._h, c2._h, c3._h if c3 is not None else None, B, rows.data_ptr(),
                                                   out.data_ptr(), tier.data_ptr(), int(threshold),
                                                   torch.cuda.current_stream(c1.device).cuda_stream))
    return tier, out
