"""The Cython EvLFU module's call surface -- cache_algo/EvLFU_C1_Cython/EvLFU.pyx:11-25 (cinit, crequest,
cload_ev_tables, cclose_ev_tables) over evlfu.hpp / EvLFU.cpp:168-232.  Same policy as EvLFU_C1 with the C++
constants flush_rate 0.4 / perfect_item_cap 1.0 (EvLFU.cpp:12-13): variant "cython" of the cache tier
(csrc/evs_cache.hip), pinned bit-exact to traces of the reference's EvLFU.cpp compiled in place
(tests/golden/cython_traces.npz)."""
from ._common import _ModuleCache

_m = _ModuleCache("evlfu")
flush_rate = 0.4            # EvLFU.cpp:12
perfect_item_cap = 1.0      # EvLFU.cpp:13


def cinit(capacity, device="cuda", engine="auto"):
    _m.init(capacity, "cython", device, engine)


def crequest(group_keys, use_gpu=False):
    """-> (arr_record_hit: list[bool] * 26, arr_emb_weights: list[list[float] * 36] * 26) -- what Cython hands back
    for vector[bool] / vector[vector[float]] (EvLFU.pyx:14-18); the caller wraps each row in a FloatTensor
    (dlrm_s_pytorch_C1.py:250-253)."""
    hit, rows = _m.request_rows(group_keys)
    return hit, rows.tolist()


def cload_ev_tables():
    """EvLFU.cpp opens its 26 table files here (load_ev_tables, :127-151); this tier reads misses from the storage
    manager's tables (bound here, or on the first request)."""
    if not getattr(_m, "_bound", False):
        _m._bind()


def cclose_ev_tables():
    """close_ev_tables (EvLFU.cpp:153-160) closes the 26 FILE handles; the cache itself stays as it is.  The storage
    manager owns the tables here: nothing to do."""


def stats():
    return _m.stats()
