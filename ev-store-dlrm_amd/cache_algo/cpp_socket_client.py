"""Mirror of the reference's cache_algo/cpp_socket_client.py: the ctypes front end of the C++
cache manager (init_ctypes_lib :63-83, cache_lookup_via_ctypes :119, request_to_cpp_cache
:129-157, print_n_reset_perfect_hit :85-87).  The socket transport (:89-117) is out of scope."""
import ctypes

import torch

from .. import _lib

N_EVTable = 26
EV_DIMENSION = 36
cache_manager_cpp = None
emb_weights_in_tensor = [None] * N_EVTable  # module-global list reused across calls, as in the reference (:18)


def init_ctypes_lib(ev_table_root=None, main_precision=32, total_size=75425, n_caching_layer=1,
                    secondary_precision=4, size_proportion="", backing="hbm", altkey_dir=None):
    """Loads libevstore_hip.so.  With ev_table_root the manager is configured here; without it the
    library reads the EVS_* environment variables on the first lookup."""
    global cache_manager_cpp
    print("Initiating ctypes cache_manager_cpp library (libevstore_hip.so) ...")
    cache_manager_cpp = _lib.lib()
    if altkey_dir is not None:
        _lib.check(cache_manager_cpp.evs_manager_set_altkey_dir(str(altkey_dir).encode()))
    if ev_table_root is not None:
        _lib.check(cache_manager_cpp.evs_manager_configure(
            n_caching_layer, main_precision, secondary_precision, total_size, size_proportion.encode(),
            str(ev_table_root).encode(), 1 if backing == "pinned" else 0))


def print_n_reset_perfect_hit():
    if cache_manager_cpp is not None:
        cache_manager_cpp.print_perfect_hit()


def establish_socket_conn():
    print("ERROR: the loopback-socket transport is not part of this build; use the ctypes path")
    exit(-1)


def cache_lookup_via_ctypes(group_rowIds):
    return cache_manager_cpp.ev_lookup((ctypes.c_int * N_EVTable)(*group_rowIds))


def request_to_cpp_cache(group_rowIds, use_gpu=False, use_socket=False, evstore_gpu_id=0):
    if use_socket:
        establish_socket_conn()
    clean_arr_floats = cache_lookup_via_ctypes(group_rowIds)
    if not clean_arr_floats:
        print("ERROR: ev_lookup failed: " + _lib.lib().evs_last_error().decode())
        exit(-1)
    flat = torch.frombuffer((ctypes.c_float * (N_EVTable * EV_DIMENSION)).from_address(
        ctypes.addressof(clean_arr_floats.contents)), dtype=torch.float32).clone()
    if use_gpu:
        flat = flat.to(torch.device("cuda:" + str(evstore_gpu_id)))  # ONE copy instead of 26 (:148-150)
    for table_idx in range(N_EVTable):
        emb_weights_in_tensor[table_idx] = flat[table_idx * EV_DIMENSION:(table_idx + 1) * EV_DIMENSION].view(1, -1)
    return emb_weights_in_tensor
