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


def test_torch_extension_builds_loads_and_wraps_the_same_library():
    """The PyTorch-ROCm C++ extension (csrc/evs_torch_ext.cpp) is built in-tree, loads without a GPU, reports the
    library's ABI version, and raises the package's EvsError with the library's message on a failed call."""
    import numpy as np
    import torch
    import evstore_dlrm_amd as E
    from evstore_dlrm_amd import _ext, _ext_build, host_cache
    if not os.path.exists(_ext_build.OUT):
        E.build()
    X = _ext.ext()
    assert X is not None and X.abi_version() == 1
    for name in ("Tables", "apply_emb", "apply_emb_interact", "interact_dot", "interact_dot_pooled", "cache_request",
                 "cache_lookup_interact", "hostcache_request_list", "slices"):
        assert hasattr(X, name), name
    # slices: independent leaf tensors over one block
    blk = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)
    a = X.slices(blk, True)
    assert len(a) == 2 and a[1].shape == (3, 4) and a[1].is_leaf and a[1].requires_grad and torch.equal(a[1].detach(), blk[1])
    a[0].detach().zero_()
    assert float(blk[0].sum()) == 0.0    # aliases, not copies
    # one host-engine request through the extension = the ctypes wrapper's answer
    tabs = [np.random.RandomState(k).rand(50, 36).astype(np.float32) for k in range(26)]
    c1 = host_cache.HostCache("evlfu", 100).set_backing(tabs)
    c2 = host_cache.HostCache("evlfu", 100).set_backing(tabs)
    ids = torch.arange(26, dtype=torch.int64).reshape(26, 1) % 50
    for _ in range(2):
        flags, ly, perfect = X.hostcache_request_list(c1._h.value, ids, 26, 36, -1, False, 0)
        hit, out = c2.request(ids.numpy().reshape(1, 26).astype(np.int32))
        assert flags == [bool(v) for v in hit[0]] and perfect == bool(hit.all())
        assert np.array_equal(torch.cat(ly).detach().numpy(), out[0])
    bad = ids.clone()
    bad[3, 0] = 50
    with pytest.raises(E.EvsError) as e:
        X.hostcache_request_list(c1._h.value, bad, 26, 36, -1, False, 0)
    assert e.value.code == E._lib.EVS_EINDEX and "out of range" in str(e.value)
