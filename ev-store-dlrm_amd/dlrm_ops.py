"""Host-side mirror of the reference's hot-path operators.

Same names, argument meaning and return conventions as the reference so the
existing inference loop can call them unchanged:

  apply_emb(lS_o, lS_i, emb_l, v_W_l)      -- DLRM_Net.apply_emb       (dlrm_s_pytorch.py:407-461)
  interact_features(x, ly, ...)            -- DLRM_Net.interact_features (dlrm_s_pytorch.py:483-516)

PyTorch is used for device memory and streams only; the arithmetic is in
libevstore_hip.so (csrc/evs_gather.hip, csrc/evs_interact.hip).
"""
import collections.abc
import ctypes as C
import os
import threading
import sys
import warnings
import weakref

import torch

from . import _ext, _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device):
    """Raw handle of torch's current stream on `device` (the private fast accessor when this torch has it:
    the public one builds a Stream object, ~4 us per call -- a third of a small-batch step)."""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else torch.device(device).index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


class _PooledList(list):
    """What apply_emb returns: a plain list of T (B,d) views plus where they live, so that interact_features can
    address the T features arithmetically instead of asking 27 tensors for their pointers and strides.
    Any mutation (a caller swapping in a cached row tensor for one table, say) drops that shortcut.
    _evs_defer: the _Deferred state when the gather behind the list has not been launched yet (see there)."""
    __slots__ = ("_evs_meta", "_evs_defer", "__weakref__")

    def _dirty(name):
        base = getattr(list, name)

        def f(self, *a, **k):
            self._evs_meta = None
            self._evs_defer = None   # (the elements still materialise on first touch: they carry the state themselves)
            return base(self, *a, **k)
        f.__name__ = name
        return f

    for _n in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove",
               "clear", "reverse", "sort"):
        locals()[_n] = _dirty(_n)
    del _n, _dirty


_feat_cache = {}

# ---- deferred pooling: the DEFAULT result of apply_emb since round 4 ------------------------------------------------------
# The reference's forward is `ly = apply_emb(...)` then `interact_features(x, ly)` (dlrm_s_pytorch.py:596-601).  Run as
# written that is two kernels with a (T,B,d) intermediate (47 us per 16 384-batch); the fused launch does both in 20.  The
# result of apply_emb is therefore a REAL list -- isinstance(ly, list), torch.cat(ly), torch.stack(ly), len, indexing all as
# before -- whose T elements are (B,d) views of an allocated but NOT YET FILLED buffer: tensors of the subclass
# _DeferredRow, which sees every torch function / Tensor method that touches one of them (__torch_function__: torch.cat reads
# a list's items at the C level, but it dispatches on their TYPE first) and launches the gather into the buffer before
# the function runs -- "materialise on first touch".  interact_features recognises the untouched list and runs the ONE
# fused kernel from the saved (lS_o, lS_i) instead; the buffer is then never written.  Shape / dtype / device / stride
# queries do not materialise.  What cannot be intercepted: code that takes an element's address WITHOUT going through
# torch (a foreign C++ extension unpacking at::Tensor, the legacy torch.utils.dlpack.to_dlpack(t); t.__dlpack__() and
# torch.from_dlpack(t) ARE seen): call apply_emb(..., lazy=False) for that -- and indices / offsets
# modified IN PLACE between apply_emb and the first use (checked: tensor version counters; raises instead of serving the
# rows of the wrong batch).  A result nobody ever touches is never computed -- the forks' --ev-lookup-only mode, which calls
# apply_emb and drops the list (dlrm_s_pytorch_C1.py:744-754), has to materialize(ly) or pass lazy=False to time anything.
# EVS_DEFER_POOLING=0 switches the default back to the eager gather.
# EVS_DEFER_POISON=1 is the debug mode of all this: every deferred buffer is filled with a signalling-NaN pattern when it is
# handed out, materialize() checks that the gather overwrote every word of it, and interact_features refuses features that
# still hold the pattern (= somebody read the buffer past torch's dispatch) instead of multiplying garbage.
DEFER_POOLING = os.environ.get("EVS_DEFER_POOLING", "1") == "1"
DEFER_POISON = os.environ.get("EVS_DEFER_POISON", "0") == "1"
POISON_WORD = 0x7FA5A5A5   # exponent all ones, quiet bit clear, mantissa non-zero: a signalling NaN no fp32 sum produces
_defer_stats = {"handed_out": 0, "recycled": 0, "recycled_unconsumed": 0, "poison_checks": 0}


def defer_stats():
    """counters of the deferred default: results handed out, buffers recycled, and results that were dropped without ever being
    computed or consumed (legal -- nobody looked -- but a timing loop that does this measures nothing)"""
    return dict(_defer_stats)


def _poisoned(t):
    """number of fp32 words of t that still hold the poison pattern (synchronises; debug mode only)"""
    with torch._C.DisableTorchFunctionSubclass():
        t = t.as_subclass(torch.Tensor) if type(t) is not torch.Tensor else t
        return int((t.contiguous().view(torch.int32) == POISON_WORD).sum().item())


def _version_of(t):
    """t's version counter, or None where torch keeps none (inference tensors: torch.inference_mode())"""
    try:
        return None if t.is_inference() else t._version
    except RuntimeError:
        return None
# (two torch internals carry it -- the storage use count that says "no view of the buffer is alive" and the guard that calls
#  a torch function without re-entering __torch_function__: a torch build without either keeps the eager default)
if not (hasattr(torch._C, "_storage_Use_Count") and hasattr(torch._C, "DisableTorchFunctionSubclass")):
    DEFER_POOLING = False


class _Deferred:
    """state behind one deferred apply_emb result"""
    __slots__ = ("lS_o", "lS_i", "ev", "buf", "done", "vers", "one", "consumed")

    def _check(self):
        for t, v in self.vers:
            # (inference tensors keep no version counter: nothing to compare, and in-place writes to them are the caller's to order)
            if v is not None and t._version != v:
                raise RuntimeError("apply_emb's indices / offsets were modified in place before its (deferred) result was first used; "
                                   "consume the result first, or call apply_emb(..., lazy=False)")

    def materialize(self):
        if not self.done:
            self._check()
            apply_emb(self.lS_o, self.lS_i, self.ev, None, out=None, lazy=False, one_index_per_bag=self.one, _into=self.buf)
            self.done = True
            self.lS_o = self.lS_i = None
            if DEFER_POISON:
                _defer_stats["poison_checks"] += 1
                left = _poisoned(self.buf)
                if left:
                    raise AssertionError("EVS_DEFER_POISON: %d words of the deferred buffer were not written by the gather" % left)


_NO_TOUCH = None   # torch functions that only read metadata: filled on first use (the descriptors need torch loaded)


def _no_touch():
    global _NO_TOUCH
    if _NO_TOUCH is None:
        T = torch.Tensor
        names = ("shape", "dtype", "device", "ndim", "is_cuda", "requires_grad", "layout", "grad_fn", "is_leaf", "names", "grad", "_version")
        fs = {getattr(T, n).__get__ for n in names if hasattr(getattr(T, n, None), "__get__")}
        fs |= {T.size, T.dim, T.stride, T.numel, T.nelement, T.is_contiguous, T.element_size, T.storage_offset, T.is_floating_point,
               T.__len__, T.get_device, T.is_pinned, T.is_shared}
        _NO_TOUCH = fs
    return _NO_TOUCH


def _touch(a, cls):
    """materialise every deferred row reachable from a torch function's arguments: positional or keyword, inside lists, tuples
    and dicts at any depth (torch.cat(tensors=ly), torch.stack(tensors=ly, dim=1), nested sequences)"""
    if type(a) is cls:
        st = getattr(a, "_evs_state", None)
        if st is not None:
            st.materialize()
    elif isinstance(a, (list, tuple)):
        for b in a:
            _touch(b, cls)
    elif isinstance(a, dict):
        for b in a.values():
            _touch(b, cls)


class _DeferredRow(torch.Tensor):
    """one (B,d) element of a deferred apply_emb result (see above)"""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func not in _no_touch():
            _touch(args, cls)
            if kwargs:
                _touch(kwargs, cls)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    def _plain(self):
        """this row as an ordinary tensor (same storage), gathered"""
        st = getattr(self, "_evs_state", None)
        if st is not None:
            st.materialize()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor)

    def __deepcopy__(self, memo):   # (Tensor.__deepcopy__ wants new_empty() to return the subclass: a copy is a plain tensor)
        t = self._plain().clone()
        memo[id(self)] = t
        return t

    def __reduce_ex__(self, proto):   # pickle / torch.save: the values, as an ordinary tensor
        return self._plain().__reduce_ex__(proto)


_defer_pool = {}   # (T, B, d, device, inference mode) -> _PoolEntry list: recycled when nobody holds the previous result any more
_defer_lock = threading.Lock()   # (two serving threads must not be handed the same free entry)


def _max_refs(rows):
    return max(sys.getrefcount(r) for r in rows)


class _PoolEntry:
    """one (T,B,d) buffer, its T row objects (creating 26 subclass views costs ~45 us -- twice the step -- so they are kept)
    and the state they share.  Three independent conditions say "free again":
      * the list handed out last has died (a weakref callback on it clears `out`),
      * no row object escaped it (every row's reference count is what it was when only this entry held it -- `rc0`,
        measured then, not a constant),
      * no view of the buffer is alive (the storage's use count is what it was then -- `uc0`)."""
    __slots__ = ("buf", "rows", "st", "uc0", "rc0", "out", "_gone")

    def __init__(self, T, B, d, device):
        self.buf = buf = torch.empty((T, B, d), dtype=torch.float32, device=device)
        self.st = st = _Deferred()
        st.buf, st.done, st.consumed = buf, True, True
        self.rows = []
        for v in buf.unbind(0):
            r = v.as_subclass(_DeferredRow)
            r._evs_state = st
            self.rows.append(r)
        del v, r
        self.uc0 = torch._C._storage_Use_Count(buf.untyped_storage()._cdata)
        self.rc0 = _max_refs(self.rows)
        self.out = None
        self._gone = self._list_died   # (one bound method, made once)

    def _list_died(self, ref):
        if self.out is ref:
            self.out = None

    def free(self):
        return self.out is None and torch._C._storage_Use_Count(self.buf.untyped_storage()._cdata) == self.uc0 \
            and _max_refs(self.rows) == self.rc0

    def hand_out(self):
        ly = _PooledList(self.rows)
        self.out = weakref.ref(ly, self._gone)
        return ly


def _deferred_result(lS_o, lS_i, ev, one_index_per_bag, B):
    with _defer_lock:
        return _deferred_result_locked(lS_o, lS_i, ev, one_index_per_bag, B)


def _deferred_result_locked(lS_o, lS_i, ev, one_index_per_bag, B):
    T, d = len(ev), ev.d
    # (a buffer made under torch.inference_mode() is an inference tensor: never handed to code outside it, and the reverse)
    key = (T, B, d, ev.device, torch.is_inference_mode_enabled())
    pool = _defer_pool.get(key)
    if pool is None:
        if len(_defer_pool) > 8:
            _defer_pool.clear()
        pool = _defer_pool[key] = []
    ent = None
    for e in pool:
        if e.free():
            ent = e
            _defer_stats["recycled"] += 1
            if not (e.st.done or e.st.consumed):
                _defer_stats["recycled_unconsumed"] += 1
                if DEFER_POISON:
                    warnings.warn("EVS_DEFER_POISON: a deferred apply_emb result was dropped without being computed or consumed")
            break
    if ent is None:
        ent = _PoolEntry(T, B, d, ev.device)
        if len(pool) < 4:
            pool.append(ent)
    buf, st = ent.buf, ent.st
    if DEFER_POISON:
        with torch._C.DisableTorchFunctionSubclass():
            buf.view(torch.int32).fill_(POISON_WORD)
    st.lS_o, st.lS_i, st.ev, st.done, st.one, st.consumed = lS_o, lS_i, ev, False, bool(one_index_per_bag), False
    ts = ([lS_o] if torch.is_tensor(lS_o) else list(lS_o)) + ([lS_i] if torch.is_tensor(lS_i) else list(lS_i))
    st.vers = [(t, _version_of(t)) for t in ts]
    ly = ent.hand_out()
    ly._evs_meta = (buf.data_ptr(), B * d, d, B, d, T)
    ly._evs_defer = st
    _defer_stats["handed_out"] += 1
    return ly


def materialize(ly):
    """force the gather behind a deferred apply_emb result (no-op on anything else); returns ly"""
    st = getattr(ly, "_evs_defer", None)
    if st is not None:
        st.materialize()
    else:
        try:
            for t in ly:
                if type(t) is _DeferredRow and t._evs_state is not None:
                    t._evs_state.materialize()
        except TypeError:
            pass
    return ly

# Round 1's opt-in form of the same idea (EVS_LAZY_POOLING=1, or apply_emb(..., lazy=True)): apply_emb returns a LazyPooled
# sequence that launches nothing; interact_features recognises it and runs the ONE fused kernel.  Indexing, iteration, list
# concatenation and this package's ext_dist.alltoall materialise the rows with the gather kernel first -- but LazyPooled is a
# Sequence, not a list: torch.cat / torch.stack on it raise TypeError, and isinstance(ly, list) is False.  The plugin contract
# says "returns a list": the DEFAULT is the deferred list above (a real list, same fusion); this stays for callers that
# asked for it by name.
LAZY_POOLING = os.environ.get("EVS_LAZY_POOLING", "0") == "1"


class LazyPooled(collections.abc.Sequence):
    """The list apply_emb returns, not yet computed."""

    def __init__(self, lS_o, lS_i, ev, v_W_l):
        self.lS_o, self.lS_i, self.ev, self.v_W_l = lS_o, lS_i, ev, v_W_l
        self._ly = None

    def materialize(self):
        if self._ly is None:
            self._ly = apply_emb(self.lS_o, self.lS_i, self.ev, self.v_W_l, lazy=False)
        return self._ly

    def __len__(self):
        return len(self.ev)

    def __getitem__(self, i):
        return self.materialize()[i]

    def __iter__(self):
        return iter(self.materialize())

    def __add__(self, other):
        return list(self.materialize()) + list(other)

    def __radd__(self, other):
        return list(other) + list(self.materialize())

    def __repr__(self):
        return "LazyPooled(%d tables, %s)" % (len(self.ev), "materialized" if self._ly is not None else "pending")


def _row_weights_c(ev, v_W_l):
    """ctypes array of per-row weight pointers (or None); tensors kept alive by the returned list."""
    if v_W_l is None or all(w is None for w in v_W_l):
        return None, None
    keep = [None if w is None else w.detach().to(ev.device, torch.float32).contiguous() for w in v_W_l]
    return (C.c_void_p * len(keep))(*[None if w is None else w.data_ptr() for w in keep]), keep


class EVTables:
    """The embedding tables of one model resident in HBM ("emb_l" for the HIP path).

    Each table is kept exactly in the reference's on-disk byte layout
    (ev-table-{k}.bin: row r at byte r*d*bits/8; script/convert_ev_to_binary.py:31-69),
    so a .bin file is loaded into HBM verbatim and fp32 / 16 / 8 / 4-bit tables share
    one gather kernel that decodes on load.
    """

    def __init__(self, raw_tables, d, codec=32, device=None):
        """raw_tables: tensors in HBM, or PINNED host tensors (the host-memory miss tier: the kernels read the
        rows over the bus; `device` then names the GPU that runs them)."""
        assert codec in (32, 16, 8, 4)
        self.d, self.codec = int(d), int(codec)
        self.row_bytes = self.d * self.codec // 8
        self.raw = []
        ptrs = []
        for t in raw_tables:
            assert t.is_cuda or t.is_pinned(), "EVTables live in HBM or in pinned host memory"
            t = t.contiguous()
            if t.dtype != torch.uint8:
                t = t.view(torch.uint8)
            t = t.reshape(-1, self.row_bytes)
            self.raw.append(t)
            if t.is_cuda:
                ptrs.append(t.data_ptr())
            else:
                p = _lib.lib().evs_host_device_pointer(t.data_ptr())
                if not p:
                    raise _lib.EvsError(_lib.EVS_EINVAL, "pinned table is not device-accessible")
                ptrs.append(p)
        on_gpu = [t.device for t in self.raw if t.is_cuda]
        self.device = torch.device(device) if device is not None else (on_gpu[0] if on_gpu else torch.device("cuda"))
        self.n_rows = [int(t.shape[0]) for t in self.raw]
        T = len(self.raw)
        self._tables_c = (C.c_void_p * T)(*ptrs)
        self._n_rows_c = (C.c_int64 * T)(*self.n_rows)
        self._ptrs = ptrs
        self._xt = None

    def ext_tables(self):
        """this model's tables as the C++ extension holds them (None without the extension)"""
        if self._xt is None:
            X = _ext.ext()
            if X is None:
                return None
            idx = self.device.index
            self._xt = X.Tables(self.raw, [int(p) for p in self._ptrs], self.d, self.codec,
                                torch.cuda.current_device() if idx is None else idx)
        return self._xt

    # ---- constructors -------------------------------------------------------------
    @classmethod
    def from_fp32(cls, weights, device="cuda"):
        """weights: list of (n_k, d) fp32 tensors/arrays, or nn.EmbeddingBag modules."""
        ws = []
        for w in weights:
            if hasattr(w, "weight"):
                w = w.weight.data
            w = torch.as_tensor(w, dtype=torch.float32).to(device)
            ws.append(w)
        return cls(ws, ws[0].shape[1], 32)

    @classmethod
    def from_bin_dir(cls, ev_path, n_tables=26, d=36, codec=32, device="cuda"):
        """Load ev-table-{1..n}.bin (the reference's storage format) straight into HBM."""
        import numpy as np
        import os
        raws = []
        for k in range(n_tables):
            p = os.path.join(ev_path, "ev-table-%d.bin" % (k + 1))
            a = np.fromfile(p, dtype=np.uint8)
            if a.size % (d * codec // 8):
                raise _lib.EvsError(_lib.EVS_EIO, "%s: size is not a multiple of the row size" % p)
            raws.append(torch.from_numpy(a).to(device))
        return cls(raws, d, codec)

    def encode(self, bits):
        """The offline encoders on the GPU (script/reduce_precision.py + convert_ev_to_binary.py): these fp32
        tables in the reference's 16 / 8 / 4-bit row format, bit-exact with the reference's Python arithmetic."""
        assert self.codec == 32 and bits in (16, 8, 4)
        out = []
        stream = _stream_ptr(self.device)
        for k, t in enumerate(self.raw):
            dst = torch.empty((self.n_rows[k], self.d * bits // 8), dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib().evs_encode_table(bits, self.n_rows[k], self.d, t.data_ptr(), dst.data_ptr(), stream))
            out.append(dst)
        return EVTables(out, self.d, bits)

    def to_bin_dir(self, out_dir):
        """Write ev-table-{1..T}.bin in the reference's storage format (script/convert_ev_to_binary.py:31-69) -- the
        bytes in HBM ARE that format, so from_bin_dir(to_bin_dir(...)) is the identity for every precision."""
        from . import converters
        return converters.tables_to_bin_dir(self, out_dir)

    def __len__(self):
        return len(self.raw)

    def fp32_view(self, k):
        assert self.codec == 32
        return self.raw[k].view(torch.float32).reshape(self.n_rows[k], self.d)


def _as_evtables(emb_l):
    if isinstance(emb_l, EVTables):
        return emb_l
    cached = getattr(emb_l, "_evs_tables", None)
    if cached is not None:
        return cached
    ev = EVTables.from_fp32(list(emb_l), device=next(iter(emb_l)).weight.device
                            if hasattr(next(iter(emb_l)), "weight") else "cuda")
    try:
        emb_l._evs_tables = ev
    except Exception:
        pass
    return ev


def B_of(lS_o):
    """batch size of a list-form offsets argument (0: not a non-empty list of 1-D tensors)"""
    try:
        return int(lS_o[0].shape[0]) if len(lS_o) and lS_o[0].dim() == 1 else 0
    except Exception:
        return 0


def apply_emb(lS_o, lS_i, emb_l, v_W_l=None, out=None, check_indices=False, lazy=None, one_index_per_bag=False, _into=None):
    """Drop-in for DLRM_Net.apply_emb (dlrm_s_pytorch.py:407-461).

    lS_o: (T,B) int64 tensor or list of T (B,) tensors -- bag START offsets.
    lS_i: (T,B) int64 tensor (Criteo collate) or list of T 1-D int64 tensors.
    emb_l: EVTables (or a list / ModuleList of nn.EmbeddingBag, converted once).
    v_W_l: None or list with None / per-ROW weight vectors (weighted pooling).
    Returns a list of T (B,d) fp32 tensors in table order; they are views of one
    (T,B,d) buffer, or of `out` = the (B,F,d) interaction tile (slot 0 is left for x).
    One HIP launch for all tables.  lazy (default: the module switch LAZY_POOLING): return a LazyPooled sequence
    and launch nothing until the rows are touched -- interact_features then runs the fused kernel instead.
    one_index_per_bag=True asserts lS_o[k] == arange(B) for every table (what collate_wrapper_criteo_offset always
    produces, dlrm_data_pytorch.py:407-408; list form: every lS_i[k] has B entries): lS_o is then not read and the
    launch is the offsets-free row gather (33.4 -> 24.9 us for the 26 Kaggle tables at B = 16 384).
    """
    ev = _as_evtables(emb_l)
    T, d = len(ev), ev.d
    dev = ev.device
    defer = lazy is None and DEFER_POOLING and not LAZY_POOLING
    if lazy is None:
        lazy = LAZY_POOLING
    if lazy and out is None and not check_indices and fused_supported(T + 1, d) and \
            (ev.codec == 32 or v_W_l is None or all(w is None for w in v_W_l)):
        return LazyPooled(lS_o, lS_i, ev, v_W_l)
    stacked_o = torch.is_tensor(lS_o)
    stacked_i = torch.is_tensor(lS_i)
    if defer and _into is None and out is None and not check_indices and fused_supported(T + 1, d) and \
            (v_W_l is None or all(w is None for w in v_W_l)):
        # the default: a real list whose elements materialise on first touch (see _DeferredRow above)
        Bd = 0
        if stacked_o and stacked_i and lS_o.is_cuda and lS_i.is_cuda and lS_o.dim() == 2 and lS_i.dim() == 2 and \
                lS_o.dtype == torch.int64 and lS_i.dtype == torch.int64 and lS_i.stride(1) == 1 and lS_o.stride(1) == 1:
            Bd = int(lS_o.shape[1])
        elif not stacked_o and not stacked_i and B_of(lS_o) > 0 and len(lS_o) == T and len(lS_i) == T and \
                all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 for t in list(lS_o) + list(lS_i)):
            Bd = B_of(lS_o)
        if Bd > 0:
            return _deferred_result(lS_o, lS_i, ev, one_index_per_bag, Bd)
    if _into is None and out is None and stacked_i and stacked_o and (v_W_l is None or all(w is None for w in v_W_l)):
        xt = ev.ext_tables()
        if xt is not None:   # the C++ extension: checks, the (T,B,d) buffer and the launch without Python in between
            X = _ext.ext()
            buf = X.apply_emb(xt, lS_o, lS_i, one_index_per_bag, check_indices)
            ly = _PooledList(X.slices(buf, False))
            B = int(buf.shape[1])
            ly._evs_meta = (buf.data_ptr(), B * d, d, B, d, T)
            ly._evs_defer = None
            return ly
    if _into is None and out is None and not stacked_i and not stacked_o and (v_W_l is None or all(w is None for w in v_W_l)) and B_of(lS_o) > 0:
        xt = ev.ext_tables()
        if xt is not None and all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 for t in list(lS_o) + list(lS_i)):
            X = _ext.ext()   # list form (the reference's random-data loader): pointer tables built in C++
            buf = X.apply_emb_list(xt, list(lS_o), list(lS_i), bool(one_index_per_bag), bool(check_indices))
            ly = _PooledList(X.slices(buf, False))
            B = int(buf.shape[1])
            ly._evs_meta = (buf.data_ptr(), B * d, d, B, d, T)
            ly._evs_defer = None
            return ly
    L = _lib.lib()
    B = int(lS_o.shape[1]) if stacked_o else int(lS_o[0].shape[0])
    if out is None:
        buf = _into if _into is not None else torch.empty((T, B, d), dtype=torch.float32, device=dev)
        out_ptr, tstride, bstride = buf.data_ptr(), B * d, d
        ly = _PooledList(buf.unbind(0)) if _into is None else _PooledList()
    else:  # (B, F, d) tile: table k -> out[:, k+1, :]
        assert out.shape == (B, T + 1, d) and out.is_contiguous() and out.dtype == torch.float32
        out_ptr, tstride, bstride = out.data_ptr() + 4 * d, d, (T + 1) * d
        ly = _PooledList(out.unbind(1)[1:])
    ly._evs_meta = (out_ptr, tstride, bstride, B, d, T)
    ly._evs_defer = None
    rw_c, _keep = _row_weights_c(ev, v_W_l)
    stream = _stream_ptr(dev)
    if stacked_i and stacked_o:
        assert lS_i.dtype == torch.int64 and lS_o.dtype == torch.int64
        assert lS_i.is_cuda and lS_o.is_cuda, "indices/offsets must already be on the GPU (dlrm_wrap does this)"
        assert lS_i.stride(1) == 1 and lS_o.stride(1) == 1
        no_off = one_index_per_bag and int(lS_i.shape[1]) == B
        rc = L.evs_embedding_bag_sum_stacked(
            T, B, d, ev.codec, ev._tables_c, ev._n_rows_c,
            lS_i.data_ptr(), lS_i.stride(0), int(lS_i.shape[1]),
            None if no_off else lS_o.data_ptr(), lS_o.stride(0), rw_c, out_ptr, tstride, bstride, stream)
    else:
        li = [lS_i[k] for k in range(T)]
        lo = [lS_o[k] for k in range(T)]
        for t in li + lo:
            assert t.dtype == torch.int64 and t.is_cuda and (t.numel() == 0 or t.stride(0) == 1)
        idx_c = (C.c_void_p * T)(*[t.data_ptr() for t in li])
        off_c = (C.c_void_p * T)(*[t.data_ptr() for t in lo])
        nnz_c = (C.c_int64 * T)(*[int(t.numel()) for t in li])
        no_off = one_index_per_bag and all(int(t.numel()) == B for t in li)
        rc = L.evs_embedding_bag_sum(T, B, d, ev.codec, ev._tables_c, ev._n_rows_c, idx_c, None if no_off else off_c, nnz_c, rw_c,
                                     out_ptr, tstride, bstride, stream)
    _lib.check(rc)
    if check_indices:
        _lib.check(L.evs_check_index_errors(stream))
    return ly


def interact_features(x, ly, arch_interaction_op="dot", arch_interaction_itself=False):
    """Drop-in for DLRM_Net.interact_features (dlrm_s_pytorch.py:483-516).

    x: (B,d) fp32, ly: list of (B,d) fp32 (any row stride; e.g. views of the (B,F,d) tile).
    Returns R: (B, d + F(F-1)/2) for "dot" (F(F+1)/2 with arch_interaction_itself),
    (B, F*d) for "cat".  Unsupported op -> sys.exit like the reference (:509-514).
    """
    if arch_interaction_op not in ("dot", "cat"):
        sys.exit("ERROR: --arch-interaction-op=" + arch_interaction_op + " is not supported")
    if isinstance(ly, LazyPooled):
        if ly._ly is None and arch_interaction_op == "dot" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 \
                and x.shape[1] == ly.ev.d and x.stride(1) == 1:
            return apply_emb_interact(x, ly.lS_o, ly.lS_i, ly.ev, ly.v_W_l, arch_interaction_itself)
        ly = ly.materialize()
    st = getattr(ly, "_evs_defer", None)
    if st is not None and not st.done:
        T = len(st.ev)
        # still the list apply_emb built, nothing touched: ONE fused launch instead of the gather + the interaction
        if arch_interaction_op == "dot" and len(ly) == T and T > 0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 \
                and x.shape[1] == st.ev.d and x.stride(1) == 1 and int(x.shape[0]) == st.buf.shape[1] \
                and getattr(ly[0], "_evs_state", None) is st and getattr(ly[-1], "_evs_state", None) is st:
            st._check()
            st.consumed = True
            return apply_emb_interact(x, st.lS_o, st.lS_i, st.ev, None, arch_interaction_itself, one_index_per_bag=st.one)
        st.materialize()
    elif st is None and isinstance(ly, (list, tuple)):
        for t in ly:   # a list the caller rebuilt or changed: its deferred elements first
            if type(t) is _DeferredRow and t._evs_state is not None:
                t._evs_state.materialize()
    B, d = x.shape
    dev = x.device
    if DEFER_POISON and isinstance(ly, (list, tuple)):
        _defer_stats["poison_checks"] += 1
        for k, t in enumerate(ly):
            if torch.is_tensor(t) and t.dtype == torch.float32 and _poisoned(t):
                raise AssertionError("EVS_DEFER_POISON: feature %d still holds the unfilled-buffer pattern: it was read past torch's dispatch" % k)
    meta = getattr(ly, "_evs_meta", None)
    pooled = None
    if meta is not None and len(ly) == meta[5] and meta[3] == B and meta[4] == d and len(ly) > 0:
        base, tstride, bstride, _, _, T = meta
        # still the list apply_emb built?  (first and last element where they were put)
        if ly[0].data_ptr() == base and ly[-1].data_ptr() == base + 4 * tstride * (T - 1):
            pooled = meta
    X = _ext.ext() if arch_interaction_op == "dot" else None
    if X is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        if pooled is not None:
            return X.interact_dot_pooled(x, pooled[0], pooled[1], pooled[2], pooled[5], bool(arch_interaction_itself))
        return X.interact_dot(x, list(ly), bool(arch_interaction_itself))
    fast = None
    if pooled is not None:
        base, tstride, bstride, _, _, T = pooled
        assert x.is_cuda and x.dtype == torch.float32 and (d == 1 or x.stride(1) == 1)
        key = (x.data_ptr(), int(x.stride(0)) if B > 1 else d, base, tstride, bstride, T)
        fast = _feat_cache.get(key)
        if fast is None:
            F = T + 1
            fast = ((C.c_void_p * F)(key[0], *[base + 4 * tstride * k for k in range(T)]),
                    (C.c_int64 * F)(key[1], *([bstride if B > 1 else d] * T)), F)
            if len(_feat_cache) > 256:
                _feat_cache.clear()
            _feat_cache[key] = fast
    if fast is not None:
        ptrs, strides, F = fast
    else:
        feats = [x] + list(ly)
        F = len(feats)
        for f in feats:
            assert f.is_cuda and f.dtype == torch.float32 and f.shape == (B, d) and (d == 1 or f.stride(1) == 1)
        ptrs = (C.c_void_p * F)(*[f.data_ptr() for f in feats])
        strides = (C.c_int64 * F)(*[int(f.stride(0)) if B > 1 else d for f in feats])
    L = _lib.lib()
    if arch_interaction_op == "dot":
        P = F * (F + 1) // 2 if arch_interaction_itself else F * (F - 1) // 2
        R = torch.empty((B, d + P), dtype=torch.float32, device=dev)
        rc = L.evs_interact_dot(B, F, d, ptrs, strides, int(bool(arch_interaction_itself)),
                                R.data_ptr(), _stream_ptr(dev))
    else:
        R = torch.empty((B, F * d), dtype=torch.float32, device=dev)
        rc = L.evs_interact_cat(B, F, d, ptrs, strides, R.data_ptr(), _stream_ptr(dev))
    _lib.check(rc)
    return R


def fused_supported(F, d):
    """Shapes the fused gather+interaction kernel is built for (csrc/evs_fused.hip)."""
    return F <= 32 and d in (16, 32, 36, 48, 64, 128)


class InteractServer:
    """The fused call R = interact_features(x, apply_emb(lS_o, lS_i, emb_l)) through a RESIDENT grid (round 6;
    include/evstore_hip.h: evs_emb_interact_serve_*): post() writes one 64-byte descriptor into a pinned-host ring and returns a
    ticket -- no launch --, the grid (started by the first post, gone after idle_us without one) runs the batch with the body
    of the launched kernel (same bits), wait() spins on the answer word.  For the latency half of the metric: one
    16 384-batch posted and waited for, and small batches back to back, without the ~12 us a launch spends outside its kernel.

    Opt-in, and with rules (the header spells them out): fp32 tables, d in {16, 32, 36, 64}, T <= 27, stacked (T, B) int64
    lS_o / lS_i on the device; the inputs of a batch are COMPLETE when it is posted (synchronise the producer's stream first);
    R may be read by anything started after wait() returned; while the grid is resident it holds the compute units, so other
    launches run when it has left (idle_us, or stop())."""

    def __init__(self, emb_l, arch_interaction_itself=False, n_blocks=0, idle_us=200):
        ev = _as_evtables(emb_l)
        assert ev.codec == 32, "the resident dispatcher serves fp32 tables"
        self.ev, self.T, self.d = ev, len(ev), ev.d
        F = self.T + 1
        self.K = self.d + (F * (F + 1) // 2 if arch_interaction_itself else F * (F - 1) // 2)
        self._h = C.c_void_p()
        with torch.cuda.device(ev.device):
            _lib.check(_lib.lib().evs_emb_interact_serve_start(C.byref(self._h), self.T, self.d, ev._tables_c, ev._n_rows_c,
                                                               int(bool(arch_interaction_itself)), int(n_blocks), int(idle_us)))
        # True: the host writes descriptors through the PCIe aperture into the lines the blocks poll (large BAR); False: block 0
        # reads a mailbox in pinned host memory and republishes (EVS_SERVE_PUBLISH=leader, or no large BAR)
        self.host_published = bool(_lib.lib().evs_emb_interact_serve_mode(self._h))
        self._keep = {}       # ticket -> the tensors of a batch in flight (kept alive until it has been waited for)
        self._x = _ext.ext()

    def post(self, x, lS_o, lS_i, out=None):
        """-> (ticket, R).  The inputs must be complete (no stream orders a post)."""
        B = int(x.shape[0])
        R = out if out is not None else torch.empty((B, self.K), dtype=torch.float32, device=self.ev.device)
        if self._x is not None:
            t = self._x.serve_post(self._h.value, x, lS_o, lS_i, R, self.T, self.d, self.K)
        else:
            assert x.is_cuda and x.dtype == torch.float32 and x.shape == (B, self.d) and x.stride(1) == 1
            assert lS_i.dtype == torch.int64 and lS_o.dtype == torch.int64 and lS_i.is_cuda and lS_o.is_cuda
            assert tuple(lS_i.shape) == (self.T, B) and tuple(lS_o.shape) == (self.T, B) and lS_i.stride(1) == 1 and lS_o.stride(1) == 1
            assert R.shape == (B, self.K) and R.is_contiguous() and R.dtype == torch.float32
            tk = C.c_uint64(0)
            _lib.check(_lib.lib().evs_emb_interact_serve_post(self._h, B, x.data_ptr(), int(x.stride(0)) if B > 1 else self.d, lS_i.data_ptr(),
                                                              int(lS_i.stride(0)), lS_o.data_ptr(), int(lS_o.stride(0)), R.data_ptr(), C.byref(tk)))
            t = tk.value
        self._keep[t] = (x, lS_o, lS_i, R)
        if len(self._keep) > 128:
            for k in [k for k in self._keep if k + 64 <= t]:
                del self._keep[k]
        return t, R

    def wait(self, ticket):
        if self._x is not None:
            self._x.serve_wait(self._h.value, int(ticket))
        else:
            _lib.check(_lib.lib().evs_emb_interact_serve_wait(self._h, int(ticket)))
        self._keep.pop(ticket, None)

    def __call__(self, x, lS_o, lS_i, out=None):
        if self._x is not None:   # one extension call: post, then spin on the answer word with the GIL released
            R = out if out is not None else torch.empty((int(x.shape[0]), self.K), dtype=torch.float32, device=self.ev.device)
            self._x.serve_run(self._h.value, x, lS_o, lS_i, R, self.T, self.d, self.K)
            return R
        t, R = self.post(x, lS_o, lS_i, out)
        self.wait(t)
        return R

    def stop(self):
        """wait for every posted batch, then send the grid home (the next post starts it again)"""
        _lib.check(_lib.lib().evs_emb_interact_serve_stop(self._h))
        self._keep.clear()

    def close(self):
        if self._h:
            _lib.lib().evs_emb_interact_serve_destroy(self._h)
            self._h = C.c_void_p()
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def apply_emb_interact(x, lS_o, lS_i, emb_l, v_W_l=None, arch_interaction_itself=False, check_indices=False,
                       out=None, one_index_per_bag=False):
    """R = interact_features(x, apply_emb(lS_o, lS_i, emb_l, v_W_l)) in ONE kernel.

    The apply_emb -> interact_features pair of DLRM_Net.sequential_forward
    (dlrm_s_pytorch.py:596-601) without the (B,F,d) intermediate: table rows are
    gathered, pooled and fed to the matrix cores in registers.  Same arguments as
    apply_emb plus x; same result as the two-call path (pooled sums bit-identical,
    dot products fp32 MFMA chains).  "dot" interaction only.
    one_index_per_bag=True asserts lS_o[k] == arange(B) for every table (what
    collate_wrapper_criteo_offset always produces, dlrm_data_pytorch.py:407-408): lS_o is then
    not read at all (stacked path only).
    """
    ev = _as_evtables(emb_l)
    T, d = len(ev), ev.d
    F = T + 1
    B = int(x.shape[0])
    dev = ev.device
    assert x.is_cuda and x.dtype == torch.float32 and x.shape == (B, d) and x.stride(1) == 1
    if not fused_supported(F, d):
        return interact_features(x, apply_emb(lS_o, lS_i, ev, v_W_l), "dot", arch_interaction_itself)
    if torch.is_tensor(lS_i) and (torch.is_tensor(lS_o) or lS_o is None) and (v_W_l is None or all(w is None for w in v_W_l)):
        xt = ev.ext_tables()
        if xt is not None:   # the C++ extension: checks, output and the launch without Python in between
            return _ext.ext().apply_emb_interact(xt, x, lS_o, lS_i, bool(arch_interaction_itself), out,
                                                 bool(one_index_per_bag), bool(check_indices))
    # genuinely multi-hot batches in list form (fp32, unweighted): the pooling kernel that gives a lane group a LOOKUP and
    # reduces through LDS (csrc/evs_gather.hip: bag_sum_flat_kernel) + the interaction kernel beat the fused general
    # loop, whose lane groups own bags (B = 16 384, ~5 indices per bag: 92 vs 133 us); same bits either way
    if out is None and ev.codec == 32 and not torch.is_tensor(lS_i) and not torch.is_tensor(lS_o) and \
            (v_W_l is None or all(w is None for w in v_W_l)) and d in (16, 32, 36, 64) and \
            all(int(o.numel()) == B for o in lS_o) and B * T < sum(int(i.numel()) for i in lS_i) <= 128 * B * T:
        return interact_features(x, apply_emb(lS_o, lS_i, ev, None, lazy=False, check_indices=check_indices), "dot", arch_interaction_itself)
    P = F * (F + 1) // 2 if arch_interaction_itself else F * (F - 1) // 2
    R = out if out is not None else torch.empty((B, d + P), dtype=torch.float32, device=dev)
    assert R.shape == (B, d + P) and R.is_contiguous() and R.dtype == torch.float32
    L = _lib.lib()
    stream = _stream_ptr(dev)
    if torch.is_tensor(lS_i) and (torch.is_tensor(lS_o) or lS_o is None):  # stacked (T,B) Criteo layout: one call, no lists
        assert lS_i.dtype == torch.int64 and lS_i.is_cuda and lS_i.stride(1) == 1
        rw_c, _keep = _row_weights_c(ev, v_W_l)
        no_off = one_index_per_bag and rw_c is None and int(lS_i.shape[1]) == B
        assert no_off or lS_o is not None, "lS_o is required unless one_index_per_bag is declared"
        if lS_o is not None:
            assert lS_o.dtype == torch.int64 and lS_o.is_cuda and lS_o.stride(1) == 1 and lS_o.shape[1] == B
        _lib.check(L.evs_emb_interact_dot_stacked(
            B, T, d, ev.codec, ev._tables_c, ev._n_rows_c, x.data_ptr(), int(x.stride(0)) if B > 1 else d,
            lS_i.data_ptr(), lS_i.stride(0), int(lS_i.shape[1]), None if no_off else lS_o.data_ptr(),
            0 if no_off else lS_o.stride(0), rw_c,
            int(bool(arch_interaction_itself)), R.data_ptr(), stream))
        if check_indices:
            _lib.check(L.evs_check_index_errors(stream))
        return R
    feats = (_lib.EvsFeature * F)()
    feats[0].src, feats[0].stride = x.data_ptr(), (int(x.stride(0)) if B > 1 else d)
    keep = []
    for k in range(T):
        i, o = lS_i[k], lS_o[k]
        assert i.dtype == torch.int64 and o.dtype == torch.int64 and i.is_cuda and o.is_cuda
        # (B + 1 offsets: EmbeddingBag's include_last_offset form, the last entry ends the last bag)
        assert (i.numel() == 0 or i.stride(0) == 1) and o.stride(0) == 1 and o.numel() in (B, B + 1)
        f = feats[k + 1]
        f.offsets_len = int(o.numel())
        # a table nobody indexes in this batch has an EMPTY index tensor whose data_ptr() is NULL -- and NULL indices
        # mean "dense feature" in the C ABI: hand over any valid address instead (nnz = 0: never dereferenced)
        f.src, f.indices, f.offsets = ev._tables_c[k], (i.data_ptr() or o.data_ptr()), o.data_ptr()
        f.nnz, f.n_rows = int(i.numel()), ev.n_rows[k]
        if v_W_l is not None and v_W_l[k] is not None:
            w = v_W_l[k].detach().to(dev, torch.float32).contiguous()
            keep.append(w)
            f.row_weights = w.data_ptr()
    _lib.check(L.evs_emb_interact_dot(B, F, d, ev.codec, feats, int(bool(arch_interaction_itself)),
                                      R.data_ptr(), stream))
    if check_indices:
        _lib.check(L.evs_check_index_errors(stream))
    return R


def apply_emb_interact_multi(xs, lS_os, lS_is, emb_l, arch_interaction_itself=False, outs=None, one_index_per_bag=False):
    """K independent batches -- lists of K x (B,d), lS_o (T,B) (or None with one_index_per_bag), lS_i (T,B) -- in ONE call:
    [apply_emb_interact(xs[k], lS_os[k], lS_is[k], emb_l) for k in range(K)], bit for bit, as ONE launch per 8 batches for
    the rows-in-registers shapes (evs_emb_interact_dot_stacked_multi: a batch's last blocks drain under the next batch's
    first ones -- the rate of a caller alternating two streams, without any stream), K single launches otherwise.
    Stacked Criteo layout, unweighted; every batch the same shape."""
    ev = _as_evtables(emb_l)
    K = len(xs)
    assert K >= 1 and len(lS_is) == K and (lS_os is None or len(lS_os) == K) and (outs is None or len(outs) == K)
    T, d = len(ev), ev.d
    F = T + 1
    if not fused_supported(F, d):
        return [apply_emb_interact(xs[k], None if lS_os is None else lS_os[k], lS_is[k], ev, None, arch_interaction_itself,
                                   out=None if outs is None else outs[k], one_index_per_bag=one_index_per_bag) for k in range(K)]
    xt = ev.ext_tables()
    if xt is not None:
        return _ext.ext().apply_emb_interact_multi(xt, list(xs), None if lS_os is None else list(lS_os), list(lS_is),
                                                   bool(arch_interaction_itself), None if outs is None else list(outs), bool(one_index_per_bag))
    B = int(xs[0].shape[0])
    P = F * (F + 1) // 2 if arch_interaction_itself else F * (F - 1) // 2
    no_off = one_index_per_bag and int(lS_is[0].shape[1]) == B
    assert no_off or lS_os is not None, "lS_o is required unless one_index_per_bag is declared"
    Rs = list(outs) if outs is not None else [torch.empty((B, d + P), dtype=torch.float32, device=ev.device) for _ in range(K)]
    for k in range(K):
        assert xs[k].is_cuda and xs[k].dtype == torch.float32 and xs[k].shape == (B, d) and xs[k].stride() == xs[0].stride()
        assert lS_is[k].is_cuda and lS_is[k].dtype == torch.int64 and lS_is[k].shape == lS_is[0].shape and lS_is[k].stride() == lS_is[0].stride()
        assert Rs[k].shape == (B, d + P) and Rs[k].is_contiguous() and Rs[k].dtype == torch.float32
        assert no_off or (lS_os[k].is_cuda and lS_os[k].dtype == torch.int64 and lS_os[k].shape == (T, B) and lS_os[k].stride() == lS_os[0].stride())
    arr = lambda ts: (C.c_void_p * K)(*[t.data_ptr() for t in ts])
    _lib.check(_lib.lib().evs_emb_interact_dot_stacked_multi(
        K, B, T, d, ev.codec, ev._tables_c, ev._n_rows_c, arr(xs), int(xs[0].stride(0)) if B > 1 else d, arr(lS_is),
        lS_is[0].stride(0), int(lS_is[0].shape[1]), None if no_off else arr(lS_os), 0 if no_off else lS_os[0].stride(0),
        int(bool(arch_interaction_itself)), arr(Rs), _stream_ptr(ev.device)))
    return Rs


_w1_cache = {}


def pad_top_layer(W1, K):
    """W1 (n1, K) -> the layout evs_emb_interact_mlp1_stacked reads: zero-padded to ((n1+15)//16*16, (K+15)//16*16), cached per
    weight TENSOR (the entry holds W1 itself: a freed weight's address, shape and version counter can all come back with
    another model's tensor through the caching allocator) and its version counter (an in-place update re-pads)."""
    key = id(W1)
    hit = _w1_cache.get(key)
    if hit is not None and hit[0] is W1 and hit[1] == W1._version and hit[2] == W1.data_ptr() and hit[3].shape[1] >= K:
        return hit[3]
    n1 = int(W1.shape[0])
    kp = (K + 15) // 16 * 16
    pad = torch.zeros(((n1 + 15) // 16 * 16, kp), dtype=torch.float32, device=W1.device)
    pad[:n1, :K] = W1.detach().to(torch.float32)
    if len(_w1_cache) > 16:
        _w1_cache.clear()
    _w1_cache[key] = (W1, W1._version, W1.data_ptr(), pad)   # W1 kept alive: id() and the address stay its own
    return pad


def apply_emb_interact_mlp1(x, lS_o, lS_i, emb_l, W1, b1, relu=True, arch_interaction_itself=False, return_R=False):
    """Z1 = act(interact_features(x, apply_emb(lS_o, lS_i, emb_l)) @ W1.T + b1) in ONE kernel: the first nn.Linear (+ ReLU) of
    the top MLP (dlrm_s_pytorch.py:601-605 with create_mlp :205-245) behind the fused gather + interaction, R staying in
    LDS.  Criteo layout only: lS_i (T,B) int64 with one index per bag (lS_o must be arange and is not read), fp32 tables,
    d in {16, 32, 36}, T <= 27.  W1: (n1, d + P) as nn.Linear.weight, b1: (n1,).  -> Z1 (B, n1) [, R (B, d + P)]."""
    ev = _as_evtables(emb_l)
    T, d = len(ev), ev.d
    F = T + 1
    B = int(x.shape[0])
    P = F * (F + 1) // 2 if arch_interaction_itself else F * (F - 1) // 2
    K = d + P
    assert ev.codec == 32 and torch.is_tensor(lS_i) and lS_i.dtype == torch.int64 and lS_i.is_cuda and lS_i.shape == (T, B) and lS_i.stride(1) == 1
    assert x.is_cuda and x.dtype == torch.float32 and x.shape == (B, d) and x.stride(1) == 1
    assert W1.shape[1] == K and b1.shape == (W1.shape[0],) and W1.is_cuda and b1.is_cuda
    n1 = int(W1.shape[0])
    w1p = pad_top_layer(W1, K)
    b1c = b1.detach().to(torch.float32).contiguous()
    Z1 = torch.empty((B, n1), dtype=torch.float32, device=ev.device)
    R = torch.empty((B, K), dtype=torch.float32, device=ev.device) if return_R else None
    _lib.check(_lib.lib().evs_emb_interact_mlp1_stacked(
        B, T, d, ev._tables_c, ev._n_rows_c, x.data_ptr(), int(x.stride(0)) if B > 1 else d, lS_i.data_ptr(), lS_i.stride(0),
        int(bool(arch_interaction_itself)), w1p.data_ptr(), int(w1p.shape[1]), b1c.data_ptr(), n1, int(bool(relu)),
        Z1.data_ptr(), R.data_ptr() if R is not None else None, _stream_ptr(ev.device)))
    return (Z1, R) if return_R else Z1
