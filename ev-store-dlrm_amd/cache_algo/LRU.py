"""LRU cache module -- call surface of the reference's cache_algo/LRU.py (init :10, request_to_lru :38)."""
from ._common import _ModuleCache

_m = _ModuleCache("lru")


def init(capacity, device="cuda", engine="auto"):
    _m.init(capacity, "python", device, engine)


def request_to_lru(group_row_ids, use_gpu=False):
    return _m.request(group_row_ids, use_gpu)
