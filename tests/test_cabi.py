"""CPU: the C-ABI library loads and exports every symbol include/evstore_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            names += re.findall(r"EVS_API\s+[\w\s\*]+?\b(\w+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_something():
    d = _declared()
    assert "evs_embedding_bag_sum" in d and "evs_interact_dot" in d and len(d) >= 6


def test_library_exports_every_declared_symbol():
    import evstore_dlrm_amd as E
    if not os.path.exists(E._lib.LIB_PATH):
        E.build()
    L = ctypes.CDLL(E._lib.LIB_PATH)
    for name in _declared():
        assert hasattr(L, name), "missing export: " + name
    assert L.evs_abi_version() == 1


def test_python_binding_covers_header():
    import evstore_dlrm_amd as E
    assert sorted(E._lib.exported_symbols()) == _declared()


def test_no_cpu_fallback_in_product():
    """The product package must never import, load or call anything under oracle/ (parity rules)."""
    pkg = os.path.join(ROOT, "ev-store-dlrm_amd")
    bad = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|liboracle|oracle\.(oracle|dlrm_cpu)|orc_[a-z_]+\(")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                for n, line in enumerate(open(os.path.join(dp, f)), 1):
                    assert not bad.search(line), (dp, f, n, line)


def test_bad_arguments_are_rejected_without_a_gpu():
    import evstore_dlrm_amd as E
    L = E._lib.lib()
    rc = L.evs_embedding_bag_sum(1, 4, 16, 7, None, None, None, None, None, None, None, 0, 0, None)
    assert rc == E._lib.EVS_EINVAL and b"codec" in L.evs_last_error()
    rc = L.evs_interact_dot(4, 0, 16, None, None, 0, None, None)
    assert rc == E._lib.EVS_EINVAL


def test_more_argument_checks_without_a_gpu():
    """Every entry point validates its arguments before it touches the device: codes and messages."""
    import ctypes as C
    import evstore_dlrm_amd as E
    L = E._lib.lib()
    EINVAL = E._lib.EVS_EINVAL
    assert L.evs_encode_table(7, 10, 36, None, None, None) == EINVAL and b"codec" in L.evs_last_error()
    assert L.evs_encode_table(4, 10, 35, None, None, None) == EINVAL           # 4-bit rows need an even dimension
    assert L.evs_encode_table(8, 0, 36, None, None, None) == 0                 # nothing to do is not an error
    assert L.evs_encode_table(8, 5, 36, None, None, None) == EINVAL and b"NULL" in L.evs_last_error()
    assert L.evs_fused_dim_supported(36) == 1 and L.evs_fused_dim_supported(20) == 0
    h = C.c_void_p()
    assert L.evs_cache_create(C.byref(h), 9, 10, 26, 36, 32, 0.3, 0.95, 1, 0) == EINVAL and b"policy" in L.evs_last_error()
    assert L.evs_cache_create(C.byref(h), 0, 0, 26, 36, 32, 0.3, 0.95, 1, 0) == EINVAL and b"capacity" in L.evs_last_error()
    assert L.evs_cache_create(C.byref(h), 0, 10, 26, 36, 5, 0.3, 0.95, 1, 0) == EINVAL and b"codec" in L.evs_last_error()
    assert L.evs_cache_create(C.byref(h), 0, 10, 65, 36, 32, 0.3, 0.95, 1, 0) == EINVAL and b"n_tables" in L.evs_last_error()
    assert L.evs_cache_request(None, 1, None, None, None, -1, None) == EINVAL
    assert L.evs_cache_lookup_batch_c1c2(None, None, 4, None, None, None, 23, None) == EINVAL
    assert L.evs_cache_lookup_batch_c1c2(None, None, 0, None, None, None, 23, None) == 0
    assert L.evs_emb_interact_dot(4, 40, 36, 32, None, 0, None, None) == EINVAL   # more than 32 features
    assert L.evs_emb_interact_dot(4, 27, 20, 32, None, 0, None, None) == EINVAL   # dimension the fused kernel is not built for
    assert not L.evs_host_device_pointer(None)
