"""MI355X-native DLRM embedding lookup + EvLFU tiered cache + feature interaction.

Drop-in for the hot path of ucare-uchicago/ev-store-dlrm (see DESIGN.md,
INTEGRATION.md).  Importable as `evstore_dlrm_amd` through the shim at the
repository root (the directory name carries a hyphen).
"""
import os as _os
import sys as _sys

from . import _lib, dlrm_ops, gpu_cache  # noqa: E402
from ._lib import EvsError, build  # noqa: E402
from .dlrm_ops import (EVTables, InteractServer, LazyPooled, apply_emb, apply_emb_interact, apply_emb_interact_mlp1, apply_emb_interact_multi, fused_supported,
                       interact_features, materialize)
from .gpu_cache import (FileTier, GpuAltKeyTier, GpuCache, lookup_batch_c1c2, lookup_batch_c1c2c3, lookup_interact_c1c2,
                        lookup_interact_c1c2c3, request_c1c2, request_c1c2c3)



def runtime_started():
    """True once this process may have initialised the HIP runtime (torch's lazy initialisation has run, or this package's
    library has been loaded): runtime knobs read from the environment can no longer be relied on to take effect."""
    t = _sys.modules.get("torch")
    return bool((t is not None and t.cuda.is_initialized()) or _lib._lib is not None)


def configure_runtime(hw_queues=8):
    """An EXPLICIT call for the integrator, before the first GPU call of the process: gives the HIP runtime
    GPU_MAX_HW_QUEUES=hw_queues unless the variable is already exported.  Why: HIP folds a process's streams onto that many
    hardware queues (4 by default) and a queue runs its commands in order; the resident exact-policy server
    (GpuCache.serve_*, the GPU engine of the batch-1 cache modules) is a kernel that STAYS on its queue, so a copy or a
    kernel of the caller's that is folded onto the same queue waits until the server goes home idle (measured through the
    plug-in loop in a process with six streams: 299 us per request against 102 with eight queues).  Importing the package
    does NOT do this (round 6): the knob belongs to the application.  Returns the value in force, or None -- with a warning
    -- when the runtime has already started and the variable was not set (the call can then no longer take effect)."""
    cur = _os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        return int(cur)
    if runtime_started():
        import warnings
        warnings.warn("evstore_dlrm_amd.configure_runtime(): the HIP runtime of this process has already started; "
                      "GPU_MAX_HW_QUEUES can no longer be set from here -- export GPU_MAX_HW_QUEUES=%d or call this before the "
                      "first GPU call if the resident cache server is to have a hardware queue of its own" % hw_queues)
        return None
    _os.environ["GPU_MAX_HW_QUEUES"] = str(int(hw_queues))
    return int(hw_queues)


__all__ = ["EvsError", "build", "configure_runtime", "runtime_started", "EVTables", "InteractServer", "LazyPooled", "apply_emb", "apply_emb_interact", "apply_emb_interact_multi", "apply_emb_interact_mlp1", "interact_features", "materialize", "fused_supported",
           "GpuCache", "FileTier", "GpuAltKeyTier", "request_c1c2", "request_c1c2c3", "lookup_batch_c1c2", "lookup_interact_c1c2",
           "lookup_batch_c1c2c3", "lookup_interact_c1c2c3"]
