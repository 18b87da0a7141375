"""ctypes binding of libevstore_hip.so (include/evstore_hip.h).

The HIP library is the product: there is NO CPU fallback.  If the shared
object is missing or cannot be loaded, every op raises (loudly).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVS_LIB_PATH") or os.path.join(_HERE, "lib", "libevstore_hip.so")  # env: developer A/B builds
_lib = None

EVS_OK, EVS_EINVAL, EVS_EHIP, EVS_EINDEX, EVS_ENOMEM, EVS_ESTATE, EVS_EIO = 0, -1, -2, -3, -4, -5, -6


class EvsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libevstore_hip: %s (code %d)" % (msg, code))
        self.code = code


def build(force=False, verbose=False):
    """Compile libevstore_hip.so for gfx950 with hipcc (works without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if force:
        subprocess.check_call(cmd + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce " + LIB_PATH)
    # the PyTorch-ROCm C++ extension over the library (csrc/evs_torch_ext.cpp -> lib/_evs_torch_ext.so), g++, host only.
    # It is the default call path but not a requirement (_ext.ext() falls back to ctypes over the same C ABI): a failed
    # extension build -- no pybind11 / g++, a torch header mismatch -- warns, removes any stale .so so that the ctypes path is
    # what runs, and leaves build() green; EVS_REQUIRE_EXT=1 keeps the failure loud.
    from . import _ext_build
    if os.path.getmtime(LIB_PATH) > (os.path.getmtime(_ext_build.OUT) if os.path.exists(_ext_build.OUT) else 0):
        force = True   # relink against the library just built
    try:
        _ext_build.build(force=force, verbose=verbose)
    except Exception as e:
        if os.environ.get("EVS_REQUIRE_EXT") == "1":
            raise
        import warnings
        warnings.warn("evstore_dlrm_amd: building the C++ extension failed (%r); the ctypes call path will be used" % (e,))
        try:
            if os.path.exists(_ext_build.OUT):
                os.remove(_ext_build.OUT)
        except OSError:
            pass


_vp, _i64, _int = C.c_void_p, C.c_int64, C.c_int
_pp = C.POINTER(C.c_void_p)
_i64p = C.POINTER(C.c_int64)

class EvsFeature(C.Structure):
    """struct evs_feature (include/evstore_hip.h)"""
    _fields_ = [("src", C.c_void_p), ("stride", C.c_int64), ("indices", C.c_void_p), ("offsets", C.c_void_p),
                ("nnz", C.c_int64), ("n_rows", C.c_int64), ("row_weights", C.c_void_p), ("offsets_len", C.c_int64)]


_PROTOS = {
    "evs_abi_version": (_int, []),
    "evs_last_error": (C.c_char_p, []),
    "evs_embedding_bag_sum": (_int, [_int, _i64, _int, _int, _pp, _i64p, _pp, _pp, _i64p, _pp, _vp, _i64, _i64, _vp]),
    "evs_embedding_bag_sum_sharded": (_int, [_int, _i64, _int, _int, _pp, _i64p, _i64p, _i64p, _pp, _pp, _i64p, _pp, _vp, _i64, _i64,
                                             _i64, _i64, _vp]),
    "evs_rowsplit_route": (_int, [_int, _i64, _int, _pp, _i64p, _i64p, _pp, _vp]),
    "evs_embedding_bag_sum_p2p": (_int, [_int, _i64, _int, _int, _pp, _i64p, _i64p, _i64p, _pp, _pp, _i64p, _pp, _vp, _i64, _i64,
                                         _i64, _i64, _vp, _vp]),
    "evs_p2p_alloc": (_int, [_pp, _i64]),
    "evs_p2p_free": (_int, [_vp]),
    "evs_p2p_ipc_export": (_int, [_vp, _vp]),
    "evs_p2p_ipc_open": (_int, [_vp, _pp]),
    "evs_p2p_ipc_close": (_int, [_vp]),
    "evs_p2p_sync": (_int, [_int, _pp, C.c_uint32, _int, _pp, C.c_uint32, _vp]),
    "evs_collate_criteo_offset": (_int, [_i64, _int, _int, _vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "evs_signal_alloc": (_int, [_pp]),
    "evs_signal_free": (_int, [_vp]),
    "evs_stream_write_value": (_int, [_vp, _vp, C.c_uint32]),
    "evs_stream_wait_value": (_int, [_vp, _vp, C.c_uint32]),
    "evs_embedding_bag_sum_stacked": (_int, [_int, _i64, _int, _int, _pp, _i64p, _vp, _i64, _i64, _vp, _i64, _pp,
                                             _vp, _i64, _i64, _vp]),
    "evs_check_index_errors": (_int, [_vp]),
    "evs_host_device_pointer": (_vp, [_vp]),
    "evs_encode_table": (_int, [_int, _i64, _int, _vp, _vp, _vp]),
    "evs_interact_dot": (_int, [_i64, _int, _int, _pp, _i64p, _int, _vp, _vp]),
    "evs_fused_dim_supported": (_int, [_int]),
    "evs_emb_interact_dot": (_int, [_i64, _int, _int, _int, C.POINTER(EvsFeature), _int, _vp, _vp]),
    "evs_emb_interact_dot_stacked": (_int, [_i64, _int, _int, _int, _pp, _i64p, _vp, _i64, _vp, _i64, _i64, _vp, _i64,
                                            _pp, _int, _vp, _vp]),
    "evs_emb_interact_serve_start": (_int, [C.POINTER(_vp), _int, _int, _pp, _i64p, _int, _int, _i64]),
    "evs_emb_interact_serve_post": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, C.POINTER(C.c_uint64)]),
    "evs_emb_interact_serve_wait": (_int, [_vp, C.c_uint64]),
    "evs_emb_interact_serve_mode": (_int, [_vp]),
    "evs_emb_interact_serve_stop": (_int, [_vp]),
    "evs_emb_interact_serve_destroy": (_int, [_vp]),
    "evs_emb_interact_dot_stacked_multi": (_int, [_int, _i64, _int, _int, _int, _pp, _i64p, _pp, _i64, _pp, _i64, _i64, _pp, _i64, _int, _pp, _vp]),
    "evs_emb_interact_mlp1_stacked": (_int, [_i64, _int, _int, _pp, _i64p, _vp, _i64, _vp, _i64, _int, _vp, _int, _vp, _int, _int, _vp, _vp, _vp]),
    "evs_cache_create": (_int, [_pp, _int, _i64, _int, _int, _int, C.c_double, C.c_double, _int, _int]),
    "evs_cache_destroy": (_int, [_vp]),
    "evs_cache_set_backing": (_int, [_vp, _pp, _i64p]),
    "evs_filetier_open": (_int, [_pp, _int, C.POINTER(C.c_char_p), _i64, _i64]),
    "evs_filetier_info": (_int, [_vp, _i64p, _pp, C.POINTER(C.c_int), _i64p]),
    "evs_filetier_fetch": (_int, [_vp, _i64, _vp, _vp, C.c_uint32]),
    "evs_filetier_close": (_int, [_vp]),
    "evs_cache_set_file_backing": (_int, [_vp, _vp]),
    "evs_cache_set_batch_policy": (_int, [_vp, _int]),
    "evs_cache_staged_rows": (_i64, [_vp]),
    "evs_cache_request": (_int, [_vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "evs_cache_serve_start": (_int, [_vp, _int, _vp, _int, _i64]),
    "evs_cache_serve_request": (_int, [_vp, _vp, _vp, C.POINTER(C.c_int)]),
    "evs_cache_serve_request_dev": (_int, [_vp, _vp, _i64, _vp, C.POINTER(C.c_int)]),
    "evs_cache_serve_request_to": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "evs_cache_serve_consumed": (_int, [_vp, _int, _vp]),
    "evs_cache_set_inline_update": (_int, [_vp, _int]),
    "evs_cache_serve_stop": (_int, [_vp]),
    "evs_cache_request_c1c2": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "evs_cache_lookup_batch_c1c2": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "evs_cache_lookup_batch_c1c2c3": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "evs_cache_lookup_interact_c1c2c3": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _int, _vp, _vp, _int, _vp]),
    "evs_aprx_batch_dump": (_i64, [_vp, _i64p, _i64, _i64p, _vp]),
    "evs_cache_lookup_interact_c1c2": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _int, _vp, _vp, _int, _vp]),
    "evs_cache_lookup_batch": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "evs_cache_lookup_interact": (_int, [_vp, _i64, _vp, _vp, _i64, _int, _vp, _vp, _vp]),
    "evs_cache_batch_stats": (_int, [_vp, _i64p, _i64p, _vp]),
    "evs_cache_batch_dump": (_i64, [_vp, _vp, _i64, _vp]),
    "evs_aprx_create": (_int, [_pp, _i64, _int]),
    "evs_aprx_destroy": (_int, [_vp]),
    "evs_aprx_set_altkeys": (_int, [_vp, _pp, _i64p]),
    "evs_aprx_stats": (_int, [_vp, _i64p, _vp]),
    "evs_aprx_apply_ops": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "evs_aprx_dump_queue": (_i64, [_vp, _i64p, _i64, _vp]),
    "evs_cache_request_c1c2c3": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _int, _vp]),
    "evs_cache_stats": (_int, [_vp, _i64p, _vp]),
    "evs_cache_reset_counters": (_int, [_vp, _vp]),
    "evs_cache_dump": (_i64, [_vp, _vp, _i64, _vp]),
    "evs_manager_configure": (_int, [_int, _int, _int, _i64, C.c_char_p, C.c_char_p, _int]),
    "evs_manager_perfect_hit": (C.c_longlong, []),
    "evs_manager_set_altkey_dir": (_int, [C.c_char_p]),
    "evs_manager_aprx_hit": (C.c_longlong, []),
    "evs_manager_tier_capacity": (C.c_longlong, [_int]),
    "ev_lookup": (C.POINTER(C.c_float), [C.POINTER(C.c_int)]),
    "get_ev_values": (C.POINTER(C.c_float), [C.POINTER(C.c_int)]),
    "print_perfect_hit": (None, []),
    "ev_lookup_based_on_list_keys": (_int, [C.POINTER(C.c_int)]),
    "test_arr": (None, [C.POINTER(C.c_int)]),
    "init_global_vars": (None, []),
    "start_server_threads": (None, []),
    "evs_interact_cat": (_int, [_i64, _int, _int, _pp, _i64p, _vp, _vp]),
    "evs_hostcache_create": (_int, [_pp, _int, _i64, _int, _int, _int, C.c_double, C.c_double, _int, _int]),
    "evs_hostcache_destroy": (_int, [_vp]),
    "evs_hostcache_set_backing": (_int, [_vp, _pp, _i64p]),
    "evs_hostcache_request": (_int, [_vp, _i64, _vp, _vp, _vp, _int]),
    "evs_hostcache_request_c1c2c3": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _int]),
    "evs_hostcache_stats": (_int, [_vp, _i64p]),
    "evs_hostcache_reset_counters": (_int, [_vp]),
    "evs_hostcache_dump": (_i64, [_vp, _vp, _i64]),
    "evs_hostaprx_create": (_int, [_pp, _i64, _int]),
    "evs_hostaprx_destroy": (_int, [_vp]),
    "evs_hostaprx_set_altkeys": (_int, [_vp, _pp, _i64p]),
    "evs_hostaprx_stats": (_int, [_vp, _i64p]),
    "evs_hostaprx_apply_ops": (_int, [_vp, _i64, _vp, _vp]),
    "evs_hostaprx_dump_queue": (_i64, [_vp, _i64p, _i64]),
}


def exported_symbols():
    return sorted(_PROTOS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libevstore_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ev-store-dlrm_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)  # AttributeError = ABI mismatch, fail loudly
            fn.restype, fn.argtypes = res, args
        if L.evs_abi_version() != 1:
            raise RuntimeError("libevstore_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise EvsError(rc, lib().evs_last_error().decode("utf-8", "replace"))
