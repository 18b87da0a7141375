"""TEST INFRASTRUCTURE -- ctypes front end of oracle/evstore_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module, and only as the checker (see the header of evstore_oracle.c).
Parity status: pinned against tests/golden/*.npz (tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


def build(force=False):
    """Compile liboracle.so (gcc) and, when /root/reference is present, oracle/_ref."""
    src = os.path.join(_HERE, "evstore_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle.so"])
    if os.path.isdir("/root/reference/mixed_precs_caching"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_embedding_bag_sum.restype = C.c_int
        L.orc_embedding_bag_sum.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, _i64p, C.c_int64,
                                            _i64p, C.c_int64, C.c_void_p, _f32p]
        for nm in ("orc_interact_dot", "orc_interact_dot_f32chain"):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = [_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, _f32p]
        L.orc_decode_u8.argtypes = [_u8p, C.c_int64, _f32p]
        L.orc_decode_u4.argtypes = [_u8p, C.c_int64, _f32p]
        L.orc_decode_u16.argtypes = [C.POINTER(C.c_uint16), C.c_int64, _f32p]
        for nm in ("orc_encode_u8_arr", "orc_encode_u16_arr", "orc_encode_u4_arr"):
            getattr(L, nm).argtypes = [C.POINTER(C.c_double), C.c_int64, _i64p]
        L.orc_evlfu_new.restype = C.c_void_p
        L.orc_evlfu_new.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_evlfu_free.argtypes = [C.c_void_p]
        L.orc_evlfu_set_tables.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.orc_evlfu_request.restype = C.c_int
        L.orc_evlfu_request.argtypes = [C.c_void_p, _i32p, _u8p, _f32p, C.c_int]
        L.orc_evlfu_dump.restype = C.c_int64
        L.orc_evlfu_dump.argtypes = [C.c_void_p, _i64p, C.c_int64]
        L.orc_evlfu_state.argtypes = [C.c_void_p, _i64p]
        L.orc_c1c2_request.restype = C.c_int
        L.orc_c1c2_request.argtypes = [C.c_void_p, C.c_void_p, _i32p, _u8p, _f32p, C.c_int]
        L.orc_aprx_new.restype = C.c_void_p
        L.orc_aprx_new.argtypes = [C.c_int64, C.POINTER(C.c_void_p), C.c_int]
        L.orc_aprx_free.argtypes = [C.c_void_p]
        L.orc_aprx_state.argtypes = [C.c_void_p, _i64p]
        L.orc_aprx_apply_ops.argtypes = [C.c_void_p, C.c_int64, _i32p, C.POINTER(C.c_uint32)]
        L.orc_aprx_dump_queue.restype = C.c_int64
        L.orc_aprx_dump_queue.argtypes = [C.c_void_p, _i64p, C.c_int64]
        L.orc_c1c2c3_request.restype = C.c_int
        L.orc_c1c2c3_request.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _i32p, _u8p, _f32p, C.c_int]
        for p in ("lru", "lfu"):
            getattr(L, "orc_%s_new" % p).restype = C.c_void_p
            getattr(L, "orc_%s_new" % p).argtypes = [C.c_int64, C.c_int, C.c_int]
            getattr(L, "orc_%s_free" % p).argtypes = [C.c_void_p]
            getattr(L, "orc_%s_set_tables" % p).argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
            getattr(L, "orc_%s_request" % p).restype = C.c_int
            getattr(L, "orc_%s_request" % p).argtypes = [C.c_void_p, _i32p, _u8p, _f32p]
            getattr(L, "orc_%s_dump" % p).restype = C.c_int64
            getattr(L, "orc_%s_dump" % p).argtypes = [C.c_void_p, _i64p, C.c_int64]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


# ---------------------------------------------------------------- codecs (a10/a11)
def decode(raw, codec, d):
    """raw: uint8/uint16 array of whole rows in the on-disk layout -> float32 (n, d)."""
    L = lib()
    if codec == 32:
        return np.ascontiguousarray(raw).view(np.float32).reshape(-1, d).copy()
    if codec == 16:
        a = np.ascontiguousarray(raw).view(np.uint16).reshape(-1)
        out = np.empty(a.size, np.float32)
        L.orc_decode_u16(_p(a, C.POINTER(C.c_uint16)), a.size, _p(out, _f32p))
        return out.reshape(-1, d)
    a = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    if codec == 8:
        out = np.empty(a.size, np.float32)
        L.orc_decode_u8(_p(a, _u8p), a.size, _p(out, _f32p))
    elif codec == 4:
        out = np.empty(a.size * 2, np.float32)
        L.orc_decode_u4(_p(a, _u8p), a.size, _p(out, _f32p))
    else:
        raise ValueError("codec must be 32, 16, 8 or 4")
    return out.reshape(-1, d)


def encode(values, codec):
    """float values -> integer codes (one per value; 4-bit codes are NOT yet packed)."""
    L = lib()
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    out = np.empty(v.size, np.int64)
    fn = {8: L.orc_encode_u8_arr, 16: L.orc_encode_u16_arr, 4: L.orc_encode_u4_arr}[codec]
    fn(_p(v, C.POINTER(C.c_double)), v.size, _p(out, _i64p))
    return out.reshape(np.shape(values))


def encode_table(W, codec):
    """fp32 table (n, d) -> raw on-disk bytes of the given codec
    (script/convert_ev_to_binary.py:31-69; 4-bit packs dim 2j in the high nibble,
    script/reduce_precision.py:321)."""
    W = np.asarray(W)
    if codec == 32:
        return np.ascontiguousarray(W, np.float32).view(np.uint8).reshape(W.shape[0], -1)
    codes = encode(W, codec)
    if codec == 16:
        return codes.astype(np.uint16).view(np.uint8).reshape(W.shape[0], -1)
    if codec == 8:
        return codes.astype(np.uint8)
    hi, lo = codes[:, 0::2], codes[:, 1::2]
    return (hi * 16 + lo).astype(np.uint8)


# ---------------------------------------------------------------- a1
def embedding_bag_sum(W, idx, off, row_weights=None, codec=32, d=None):
    """One table. W: fp32 (n,d), or raw codec bytes (n, d*bits/8) with d given."""
    L = lib()
    W = np.ascontiguousarray(W)
    if codec == 32:
        W = W.astype(np.float32, copy=False)
        n, d = W.shape
    else:
        n = W.shape[0]
        assert d is not None
    idx = np.ascontiguousarray(idx, np.int64)
    off = np.ascontiguousarray(off, np.int64)
    out = np.empty((off.size, d), np.float32)
    rw = None
    if row_weights is not None:
        rw = np.ascontiguousarray(row_weights, np.float32)
    rc = L.orc_embedding_bag_sum(W.ctypes.data, codec, n, d, _p(idx, _i64p), idx.size, _p(off, _i64p),
                                 off.size, rw.ctypes.data if rw is not None else None, _p(out, _f32p))
    if rc:
        raise IndexError("orc_embedding_bag_sum rc=%d" % rc)
    return out


def apply_emb(lS_o, lS_i, tables, v_W_l=None, codec=32, d=None):
    """dlrm_s_pytorch.py:407-461 over numpy inputs: returns list of (B,d) arrays."""
    ly = []
    for k in range(len(tables)):
        w = None if v_W_l is None else v_W_l[k]
        ly.append(embedding_bag_sum(tables[k], lS_i[k], lS_o[k], w, codec, d))
    return ly


# ---------------------------------------------------------------- a3
def interact_features(x, ly, itself=False, f32chain=False):
    L = lib()
    T = np.ascontiguousarray(np.stack([x] + list(ly), axis=1), np.float32)  # (B,F,d)
    B, F, d = T.shape
    P = F * (F + 1) // 2 if itself else F * (F - 1) // 2
    R = np.empty((B, d + P), np.float32)
    fn = L.orc_interact_dot_f32chain if f32chain else L.orc_interact_dot
    fn(_p(T, _f32p), B, F, d, int(itself), _p(R, _f32p))
    return R


# ---------------------------------------------------------------- a6/a7 policies
class _Policy:
    _prefix = None

    def __init__(self, handle, tables, n_tables, dim):
        self._h = handle
        self.n_tables, self.dim = n_tables, dim
        self._tables = [np.ascontiguousarray(t, np.float32) for t in tables]
        arr = (C.c_void_p * n_tables)(*[t.ctypes.data for t in self._tables])
        getattr(lib(), "orc_%s_set_tables" % self._prefix)(self._h, arr)
        self._hit = np.zeros(n_tables, np.uint8)
        self._out = np.zeros((n_tables, dim), np.float32)
        self._rows = np.zeros(n_tables, np.int32)

    def __del__(self):
        try:
            getattr(lib(), "orc_%s_free" % self._prefix)(self._h)
        except Exception:
            pass


# (flush_rate, perfect_item_cap, flush_extra, perfect_mode) of the three EvLFU builds in the reference
EVLFU_VARIANTS = {"python": (0.3, 0.95, 1, 0), "cpp": (0.3, 0.95, 0, 2), "cython": (0.4, 1.0, 1, 1)}


class EvLFU(_Policy):
    """cache_algo/EvLFU_C1.py. variant='python' (0.3/0.95, flush n+1), 'cpp' (0.3/0.95, flush n;
    mixed_precs_caching/evlfu_8.cpp:252-300), 'cython' (0.4/1.0; EvLFU_C1_Cython/EvLFU.cpp:12-13)."""
    _prefix = "evlfu"

    def __init__(self, cap, tables, dim=36, variant="python"):
        fr, pc, ex, pm = EVLFU_VARIANTS[variant]
        h = lib().orc_evlfu_new(cap, len(tables), dim, fr, pc, ex, pm)
        super().__init__(h, tables, len(tables), dim)

    def request(self, rows, approx_thres=-1):
        self._rows[:] = rows
        rc = lib().orc_evlfu_request(self._h, _p(self._rows, _i32p), _p(self._hit, _u8p),
                                     _p(self._out, _f32p), approx_thres)
        if rc < 0:
            raise RuntimeError("orc_evlfu_request rc=%d" % rc)
        return self._hit.astype(bool), self._out

    def dump(self):
        n = lib().orc_evlfu_dump(self._h, None, 0)
        out = np.zeros((n, 3), np.int64)
        lib().orc_evlfu_dump(self._h, _p(out, _i64p), n)
        return out

    def state(self):
        s = np.zeros(5, np.int64)
        lib().orc_evlfu_state(self._h, _p(s, _i64p))
        return dict(min_c1=int(s[0]), n_perfect=int(s[1]), size=int(s[2]), n_flush=int(s[3]),
                    n_evict=int(s[4]))


class LRU(_Policy):
    _prefix = "lru"

    def __init__(self, cap, tables, dim=36):
        super().__init__(lib().orc_lru_new(cap, len(tables), dim), tables, len(tables), dim)

    def request(self, rows):
        self._rows[:] = rows
        lib().orc_lru_request(self._h, _p(self._rows, _i32p), _p(self._hit, _u8p), _p(self._out, _f32p))
        return self._hit.astype(bool), self._out

    def dump(self):
        n = lib().orc_lru_dump(self._h, None, 0)
        out = np.zeros((n, 2), np.int64)
        lib().orc_lru_dump(self._h, _p(out, _i64p), n)
        return out


class LFU(_Policy):
    _prefix = "lfu"

    def __init__(self, cap, tables, dim=36):
        super().__init__(lib().orc_lfu_new(cap, len(tables), dim), tables, len(tables), dim)

    def request(self, rows):
        self._rows[:] = rows
        rc = lib().orc_lfu_request(self._h, _p(self._rows, _i32p), _p(self._hit, _u8p), _p(self._out, _f32p))
        if rc < 0:
            raise RuntimeError("orc_lfu_request rc=%d" % rc)
        return self._hit.astype(bool), self._out

    def dump(self):
        n = lib().orc_lfu_dump(self._h, None, 0)
        out = np.zeros((n, 3), np.int64)
        lib().orc_lfu_dump(self._h, _p(out, _i64p), n)
        return out


# ---------------------------------------------------------------- a13 miss-path readers
def read_row(bin_dir, table1, row, codec=32, d=36):
    """emb_storage/file_read.py:27-33: seek(bytes_per_row*row); read(bytes_per_row)."""
    bpr = d * codec // 8
    with open(os.path.join(bin_dir, "ev-table-%d.bin" % table1), "rb") as f:
        f.seek(bpr * row)
        raw = np.frombuffer(f.read(bpr), np.uint8)
    return decode(raw, codec, d)[0]


def kaggle_tables(n_rows, seed, d=36):
    """Synthetic tables drawn like create_emb (dlrm_s_pytorch.py:279-283): U(-sqrt(1/n), sqrt(1/n))."""
    rs = np.random.RandomState(seed)
    return [rs.uniform(-np.sqrt(1.0 / n), np.sqrt(1.0 / n), size=(n, d)).astype(np.float32)
            for n in n_rows]


def ref_tier_capacities(n_layer, main, secondary, total_size, proportion=""):
    """Entries per tier as the reference's constructors compute them (sizes are in fp32-row equivalents):
    cache_manager.cpp:31-53 (cacheSize = TOTAL_SIZE / n for 1 or 2 layers; main 32 -> cacheSize, 16 -> x2, 4 -> x8, and the
    8-bit main tier gets TOTAL_SIZE itself), evlfu_8.cpp:57-97 (x4 for its own rows; with a C2: total/2*4 and total/2*8;
    three tiers: the "a-b-c" proportion, x4 / x8 / x36), evlfu_32.cpp:99-105 and evlfu_16.cpp:93-97 (C2 = cap_C1 x2 / x4 /
    x8 for a 16 / 8 / 4-bit secondary tier from a 32-bit main, x2 / x4 from a 16-bit main).  NB an 8-bit SECONDARY tier is
    built through EVLFU_8BIT's constructor, which multiplies by 4 again (evlfu_8.cpp:93): (TOTAL/2)*16 entries, not *4.
    -> (cap_c1, cap_c2, cap_c3) with 0 for an absent tier."""
    if main == 8:
        if n_layer == 3 and proportion:
            p1, p2, p3 = [int(v) for v in proportion.split("-")]
            return (p1 * total_size // 100) * 4, (p2 * total_size // 100) * 8, (p3 * total_size // 100) * 36
        if n_layer == 3:
            return total_size // 3 * 4, total_size // 3 * 8, total_size // 3 * 36
        if n_layer == 2:
            return total_size // 2 * 4, total_size // 2 * 8, 0
        return total_size * 4, 0, 0
    cs = total_size // 2 if n_layer == 2 else total_size
    c1 = cs * (32 // main)
    if n_layer < 2:
        return c1, 0, 0
    if main == 32:
        c2 = {16: c1 * 2, 8: c1 * 4 * 4, 4: c1 * 8}[secondary]
    else:   # main 16
        c2 = {8: c1 * 2 * 4, 4: c1 * 4}[secondary]
    return c1, c2, 0


class C1C2:
    """Two-tier request (mixed_precs_caching/evlfu_8.cpp:669-796).  tables_c1 / tables_c2: the rows
    decoded at each tier's precision (fp32 arrays)."""

    def __init__(self, cap_c1, cap_c2, tables_c1, tables_c2, dim=36, threshold=23):
        self.c1 = EvLFU(cap_c1, tables_c1, dim, "cpp")
        self.c2 = EvLFU(cap_c2, tables_c2, dim, "cpp")
        self.T, self.dim, self.threshold = len(tables_c1), dim, threshold
        self._rows = np.zeros(self.T, np.int32)
        self._tier = np.zeros(self.T, np.uint8)
        self._out = np.zeros((self.T, dim), np.float32)

    def request(self, rows):
        self._rows[:] = rows
        rc = lib().orc_c1c2_request(self.c1._h, self.c2._h, _p(self._rows, _i32p), _p(self._tier, _u8p),
                                    _p(self._out, _f32p), self.threshold)
        if rc < 0:
            raise RuntimeError("orc_c1c2_request rc=%d" % rc)
        return self._tier.copy(), self._out, rc


class AltKeyTier:
    """The alt-key tier alone, through APRX_EV's public single-key methods (aprx_embedding.cpp:278-288,341-350,390-411);
    PINNED to the reference driven single-threaded (tests/golden/aprx_ops.npz).  ops: (n,3) int32 rows of
    (op, table_1based, row), op 0 insert | 1 lookup | 2 set recency flag | 3 evict one."""

    def __init__(self, cap, alt_tables):
        self._alt = [np.ascontiguousarray(a, np.uint32) for a in alt_tables]
        arr = (C.c_void_p * len(self._alt))(*[a.ctypes.data for a in self._alt])
        self._h = lib().orc_aprx_new(cap, arr, len(self._alt))
        if not self._h:
            raise ValueError("alt-key tier capacity must be >= 50")

    def apply(self, ops):
        ops = np.ascontiguousarray(ops, np.int32).reshape(-1, 3)
        res = np.zeros(len(ops), np.uint32)
        lib().orc_aprx_apply_ops(self._h, len(ops), _p(ops, _i32p), _p(res, C.POINTER(C.c_uint32)))
        return res

    def queue(self):
        n = lib().orc_aprx_dump_queue(self._h, None, 0)
        out = np.zeros((n, 2), np.int64)
        lib().orc_aprx_dump_queue(self._h, _p(out, _i64p), n)
        return out

    def state(self):
        s = np.zeros(4, np.int64)
        lib().orc_aprx_state(self._h, _p(s, _i64p))
        return dict(size=int(s[0]), n_hit=int(s[1]), n_pending=int(s[2]), error=int(s[3]))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_aprx_free(self._h)
            self._h = None


class C1C2C3(C1C2):
    """C1 + C2 + the alt-key tier (deterministic re-specification of evlfu_8.cpp:492-667; parity unpinned:
    the reference's tier is asynchronous).  alt_tables: per table a uint32 array, alt_key[row] = alt_row*100 + alt_table."""

    def __init__(self, cap_c1, cap_c2, cap_c3, tables_c1, tables_c2, alt_tables, dim=36, threshold=23):
        super().__init__(cap_c1, cap_c2, tables_c1, tables_c2, dim, threshold)
        self._alt = [np.ascontiguousarray(a, np.uint32) for a in alt_tables]
        arr = (C.c_void_p * self.T)(*[a.ctypes.data for a in self._alt])
        self.c3 = lib().orc_aprx_new(cap_c3, arr, self.T)
        if not self.c3:
            raise ValueError("alt-key tier capacity must be >= 50 (aprx_embedding.cpp:33 asserts cap_C3 >= IO_JOB_Q_SIZE)")

    def request(self, rows):
        self._rows[:] = rows
        rc = lib().orc_c1c2c3_request(self.c1._h, self.c2._h, self.c3, _p(self._rows, _i32p), _p(self._tier, _u8p),
                                      _p(self._out, _f32p), self.threshold)
        if rc < 0:
            raise RuntimeError("orc_c1c2c3_request rc=%d" % rc)
        return self._tier.copy(), self._out, rc

    def c3_state(self):
        s = np.zeros(4, np.int64)
        lib().orc_aprx_state(self.c3, _p(s, _i64p))
        return dict(size=int(s[0]), n_hit=int(s[1]), n_pending=int(s[2]), error=int(s[3]))
