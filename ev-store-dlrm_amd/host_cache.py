"""Host engine of the exact (one-request-at-a-time) cache policies -- Python handle over evs_hostcache_* / evs_hostaprx_*
(csrc/evs_hostcache.hip, include/evstore_hip.h).

Same policies, constants and call shapes as gpu_cache.GpuCache / GpuAltKeyTier / request_c1c2c3, but every buffer is host
memory (numpy arrays, or CPU torch tensors through their numpy views) and nothing touches the GPU: this is where the
reference's batch-1 EVStore loop (dlrm_s_pytorch_C1.py:236-239) runs at a few microseconds per request.  The batched
snapshot lookups stay on the GPU tier (gpu_cache.py).
"""
import ctypes as C

import numpy as np

from . import _lib
from .gpu_cache import EVLFU_VARIANTS, POLICY


def _np(a, dtype):
    """numpy view of a numpy array or CPU torch tensor, C-contiguous, of `dtype` (no copy when already so)."""
    if hasattr(a, "numpy") and not isinstance(a, np.ndarray):
        a = a.detach().numpy() if hasattr(a, "detach") else a.numpy()
    return np.ascontiguousarray(a, dtype=dtype)


class HostCache:
    def __init__(self, policy, capacity, n_tables=26, dim=36, codec=32, variant="python"):
        self.policy, self.capacity, self.n_tables, self.dim, self.codec = policy, int(capacity), n_tables, dim, codec
        fr, pc, ex, pm = EVLFU_VARIANTS[variant]
        h = C.c_void_p()
        _lib.check(_lib.lib().evs_hostcache_create(C.byref(h), POLICY[policy], self.capacity, n_tables, dim, codec, fr, pc, ex, pm))
        self._h = h
        self._backing = None

    def __del__(self):
        try:
            if self._h:
                _lib.lib().evs_hostcache_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_backing(self, tables):
        """tables: per table a host array holding the table in the cache's codec (numpy array / np.memmap of the .bin
        file / CPU or pinned torch tensor), row r at byte r * dim * codec / 8."""
        raws = tables.raw if hasattr(tables, "raw") else list(tables)
        if len(raws) != self.n_tables:
            raise ValueError("%d tables for a cache of %d" % (len(raws), self.n_tables))
        rb = self.dim * self.codec // 8
        keep, ptrs, rows = [], [], []
        for t in raws:
            if hasattr(t, "is_cuda"):
                if t.is_cuda:
                    raise ValueError("the host engine reads its miss tier from HOST memory")
                t = t.contiguous().numpy()
            a = t if isinstance(t, np.ndarray) and t.flags["C_CONTIGUOUS"] else np.ascontiguousarray(t)
            keep.append(a)
            ptrs.append(a.ctypes.data)
            rows.append(a.nbytes // rb)
        self._backing = keep
        _lib.check(_lib.lib().evs_hostcache_set_backing(self._h, (C.c_void_p * self.n_tables)(*ptrs), (C.c_int64 * self.n_tables)(*rows)))
        return self

    def request(self, rows, approx_thres=-1, out=None, hit=None):
        """rows: (B, n_tables) int32.  -> (hit (B,T) uint8, out (B,T,dim) float32), numpy."""
        rows = _np(rows, np.int32).reshape(-1, self.n_tables)
        B = rows.shape[0]
        if out is None:
            out = np.empty((B, self.n_tables, self.dim), np.float32)
        if hit is None:
            hit = np.empty((B, self.n_tables), np.uint8)
        if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size == B * self.n_tables * self.dim
                and isinstance(hit, np.ndarray) and hit.dtype == np.uint8 and hit.flags["C_CONTIGUOUS"] and hit.size == B * self.n_tables):
            raise ValueError("out must be a C-contiguous float32 array of B * n_tables * dim elements, hit a uint8 one of B * n_tables")
        _lib.check(_lib.lib().evs_hostcache_request(self._h, B, rows.ctypes.data, out.ctypes.data, hit.ctypes.data, int(approx_thres)))
        return hit, out

    def stats(self):
        s = (C.c_int64 * 8)()
        _lib.check(_lib.lib().evs_hostcache_stats(self._h, s))
        keys = ("min_c1", "n_perfect", "size", "n_flush", "n_evict", "n_requests", "n_perfect_hits", "n_hits")
        return dict(zip(keys, [int(v) for v in s]))

    def reset_counters(self):
        _lib.check(_lib.lib().evs_hostcache_reset_counters(self._h))

    def dump(self):
        """Resident keys in list order: rows of (bucket | frequency | 0, table_1based, row)."""
        n = _lib.lib().evs_hostcache_dump(self._h, None, 0)
        if n < 0:
            _lib.check(int(n))
        out = np.zeros((max(n, 1), 3), np.int64)
        _lib.lib().evs_hostcache_dump(self._h, out.ctypes.data, n)
        return out[:n]


class HostAltKeyTier:
    """C3 on the host (evs_hostaprx_*): key -> alt key with the second-chance FIFO; alt_tables: per table a uint32 array
    with alt_key[row] = alt_row * 100 + alt_table_1based (native byte order)."""

    def __init__(self, capacity, alt_tables):
        self.n_tables = len(alt_tables)
        h = C.c_void_p()
        _lib.check(_lib.lib().evs_hostaprx_create(C.byref(h), int(capacity), self.n_tables))
        self._h = h
        self._alt = [_np(t, np.uint32) for t in alt_tables]
        ptrs = (C.c_void_p * self.n_tables)(*[a.ctypes.data for a in self._alt])
        rows = (C.c_int64 * self.n_tables)(*[int(a.size) for a in self._alt])
        _lib.check(_lib.lib().evs_hostaprx_set_altkeys(self._h, ptrs, rows))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().evs_hostaprx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def stats(self):
        s = (C.c_int64 * 4)()
        _lib.check(_lib.lib().evs_hostaprx_stats(self._h, s))
        return dict(size=int(s[0]), n_hit=int(s[1]), n_pending=int(s[2]), error=int(s[3]))

    def apply_ops(self, ops):
        ops = _np(ops, np.int32).reshape(-1, 3)
        res = np.zeros((ops.shape[0],), np.uint32)
        _lib.check(_lib.lib().evs_hostaprx_apply_ops(self._h, ops.shape[0], ops.ctypes.data, res.ctypes.data))
        return res.astype(np.int64)

    def queue(self):
        n = int(_lib.lib().evs_hostaprx_dump_queue(self._h, None, 0))
        if n < 0:
            _lib.check(n)
        out = np.zeros((max(n, 1), 2), np.int64)
        _lib.lib().evs_hostaprx_dump_queue(self._h, out.ctypes.data_as(C.POINTER(C.c_int64)), n)
        return out[:n]


def request_c1c2c3(c1, c2, c3, rows, high_agghit_threshold=23, out=None, tier=None):
    """request_to_c1_c2 (c3 None) / request_to_c1_c2_c3 on the host: -> (tier (B,T) uint8: 1 C1, 2 C2, 3 alt key, 0 miss;
    out (B,T,dim) float32)."""
    rows = _np(rows, np.int32).reshape(-1, c1.n_tables)
    B = rows.shape[0]
    if out is None:
        out = np.empty((B, c1.n_tables, c1.dim), np.float32)
    if tier is None:
        tier = np.empty((B, c1.n_tables), np.uint8)
    if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size == B * c1.n_tables * c1.dim
            and isinstance(tier, np.ndarray) and tier.dtype == np.uint8 and tier.flags["C_CONTIGUOUS"] and tier.size == B * c1.n_tables):
        raise ValueError("out must be a C-contiguous float32 array of B * n_tables * dim elements, tier a uint8 one of B * n_tables")
    _lib.check(_lib.lib().evs_hostcache_request_c1c2c3(c1._h, c2._h, c3._h if c3 is not None else None, B, rows.ctypes.data,
                                                       out.ctypes.data, tier.ctypes.data, int(high_agghit_threshold)))
    return tier, out


def request_c1c2(c1, c2, rows, high_agghit_threshold=23, out=None, tier=None):
    return request_c1c2c3(c1, c2, None, rows, high_agghit_threshold, out, tier)
