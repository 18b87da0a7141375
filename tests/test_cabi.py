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
