"""EvLFU cache module -- same call surface as the reference's cache_algo/EvLFU_C1.py
(init :21, request_to_ev_lfu :97), state held by the host engine (csrc/evs_hostcache.hip) or in HBM (csrc/evs_cache.hip)."""
from ._common import _ModuleCache

_m = _ModuleCache("evlfu")
flush_rate_C1 = 0.3          # EvLFU_C1.py:18
perfect_item_cap_C1 = 0.95   # EvLFU_C1.py:19


def init(capacity, variant="python", device="cuda", engine="auto"):
    """engine: "auto" (host engine unless the tables live in HBM only), "host", "gpu" -- see _common.py"""
    _m.init(capacity, variant, device, engine)


def request_to_ev_lfu(group_row_ids, use_gpu=False, approx_emb_thres=-1, ev_dim=36):
    """-> (arr_record_hit: list[bool]*26, arr_emb_weights: list[Tensor(1,36)]*26)"""
    return _m.request(group_row_ids, use_gpu, approx_emb_thres)


def stats():
    return _m.stats()
