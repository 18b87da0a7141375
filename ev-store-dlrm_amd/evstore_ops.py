"""apply_emb_evstore -- mirror of dlrm_s_pytorch_C1.py:227-275 and of the _C1_C2 / _C1_C2_C3 forks' version
(dlrm_s_pytorch_C1_C2.py:227-275): ONE if-chain with every branch the forks have.

Module globals `cache_algo`, `perfect_hit` and (C1_C2 forks) `evstore_gpu_id` as in the reference.  The lookup itself
runs in the GPU cache tier (cache_algo/*.py -> csrc/evs_cache.hip) or, for "cpp_algo", in the cache manager of
libevstore_hip.so through the ctypes client (cache_algo/cpp_socket_client.py -> ev_lookup)."""
import torch

from .cache_algo import EvLFU, EvLFU_C1, LFU, LRU, cpp_socket_client
from .emb_storage import storage_manager

cache_algo = "evlfu"
perfect_hit = 0
evstore_gpu_id = 0   # dlrm_s_pytorch_C1_C2.py:1267 (--evstore-gpu-id)
_FAST = {"evlfu": EvLFU_C1, "lru": LRU, "lfu": LFU}


def apply_emb_evstore(lS_o, lS_i, emb_l, v_W_l, use_gpu=False, use_emb_cache=False, approx_emb_threshold=-1):
    """Takes element 0 of each table's index row (batch size 1, dlrm_s_pytorch_C1.py:236-239) -> 26 row
    ids -> cache / storage -> list of 26 Tensor(1,36).  lS_o, emb_l, v_W_l are ignored as in the reference."""
    global perfect_hit
    if use_emb_cache and cache_algo in _FAST:
        # host engine + C++ extension: ids, policy and the 26 tensors in one call (same results as the generic path below);
        # a device lS_i comes over through pinned staging inside that call instead of the reference's lS_i.cpu()
        mod = _FAST[cache_algo]
        r = mod._m.request_from_index_rows(lS_i, use_gpu, approx_emb_threshold if cache_algo == "evlfu" else -1)
        if r is not None:
            if r[2]:
                perfect_hit += 1
            return r[1]
    if use_gpu:
        lS_i = lS_i.cpu().data
    group_rowIds = [int(sparse_index[0]) for sparse_index in lS_i.numpy()]
    if use_emb_cache:
        if cache_algo == "evlfu":
            aggHitMissRecord, ly = EvLFU_C1.request_to_ev_lfu(group_rowIds, use_gpu, approx_emb_threshold)
        elif cache_algo == "lru":
            aggHitMissRecord, ly = LRU.request_to_lru(group_rowIds, use_gpu)
        elif cache_algo == "lfu":
            aggHitMissRecord, ly = LFU.request_to_lfu(group_rowIds, use_gpu)
        elif cache_algo == "evlfu_cython":
            # dlrm_s_pytorch_C1.py:250-253: rows come back as 26 float lists, each wrapped in a CPU FloatTensor
            aggHitMissRecord, tmp = EvLFU.crequest(group_rowIds, use_gpu)
            ly = [torch.FloatTensor([tmp[i]]) for i in range(len(tmp))]
        elif cache_algo == "cpp_algo":
            # dlrm_s_pytorch_C1_C2.py:247-249: the ctypes boundary; the perfect-hit count lives in the library
            # (print_perfect_hit), the forks leave aggHitMissRecord empty and count nothing here
            aggHitMissRecord = None
            ly = cpp_socket_client.request_to_cpp_cache(group_rowIds, use_gpu, False, evstore_gpu_id)
        elif cache_algo == "cpp_algo_socket":
            aggHitMissRecord = None
            ly = cpp_socket_client.request_to_cpp_cache(group_rowIds, use_gpu, True)  # out of scope: prints + exits
        else:
            print("ERROR: This algorithm is not yet supported! " + str(cache_algo))
            exit(-1)
        if aggHitMissRecord is not None and all(aggHitMissRecord):
            perfect_hit += 1
    else:
        _, ly = storage_manager.request_to_emb_storage(group_rowIds, use_gpu)
    return ly
