"""Loader of lib/_evs_torch_ext.so, the PyTorch-ROCm C++ extension over libevstore_hip.so (csrc/evs_torch_ext.cpp).

The extension is the DEFAULT call path of apply_emb / interact_features / apply_emb_interact, the cache tier's
request / lookup_interact and the batch-1 EVStore request; the ctypes binding (_lib.py) reaches the SAME extern "C" entry
points of the SAME library and stays for everything else (and as the A/B: EVS_NO_EXT=1).  Both need libevstore_hip.so:
there is no CPU fallback behind either."""
import importlib.util
import os

from . import _ext_build, _lib

_mod = None
_tried = False


def ext():
    """the extension module, or None when it is not built / switched off / a developer build of the library is loaded"""
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if os.environ.get("EVS_NO_EXT", "0") == "1" or os.environ.get("EVS_LIB_PATH"):
        return None
    if not os.path.exists(_ext_build.OUT):
        return None
    _lib.lib()            # the library first: the extension resolves its symbols against the same file
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location("_evs_torch_ext", _ext_build.OUT)
    m = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(m)
    except (ImportError, OSError) as e:   # e.g. built against another torch: the ctypes path reaches the same library
        import warnings
        warnings.warn("_evs_torch_ext.so did not load (%s): calling libevstore_hip.so through ctypes; rebuild with "
                      "__graft_entry__.build()" % (e,))
        return None
    if m.abi_version() != 1:
        raise RuntimeError("_evs_torch_ext.so was built against another libevstore_hip ABI")
    m.set_error_class(_lib.EvsError)
    _mod = m
    return _mod
