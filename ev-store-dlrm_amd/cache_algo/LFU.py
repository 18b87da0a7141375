"""LFU cache module -- call surface of the reference's cache_algo/LFU.py (init :12, request_to_lfu :69)."""
from ._common import _ModuleCache

_m = _ModuleCache("lfu")


def init(capacity, device="cuda", engine="auto"):
    _m.init(capacity, "python", device, engine)


def request_to_lfu(group_row_ids, use_gpu=False):
    return _m.request(group_row_ids, use_gpu)
