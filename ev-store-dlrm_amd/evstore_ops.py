"""apply_emb_evstore -- mirror of dlrm_s_pytorch_C1.py:227-275 (and the _C1_C2/_C1_C2_C3 forks).

Module globals `cache_algo` and `perfect_hit` as in the reference.  The lookup itself runs in the
GPU cache tier (cache_algo/*.py -> csrc/evs_cache.hip)."""
from .cache_algo import EvLFU_C1, LRU, LFU
from .emb_storage import storage_manager

cache_algo = "evlfu"
perfect_hit = 0


def apply_emb_evstore(lS_o, lS_i, emb_l, v_W_l, use_gpu=False, use_emb_cache=False, approx_emb_threshold=-1):
    """Takes element 0 of each table's index row (batch size 1, dlrm_s_pytorch_C1.py:236-239) -> 26 row
    ids -> cache / storage -> list of 26 Tensor(1,36).  lS_o, emb_l, v_W_l are ignored as in the reference."""
    global perfect_hit
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
        else:
            print("ERROR: This algorithm is not yet supported! " + str(cache_algo))
            exit(-1)
        if all(aggHitMissRecord):
            perfect_hit += 1
    else:
        _, ly = storage_manager.request_to_emb_storage(group_rowIds, use_gpu)
    return ly
