"""MI355X-native DLRM embedding lookup + EvLFU tiered cache + feature interaction.

Drop-in for the hot path of ucare-uchicago/ev-store-dlrm (see DESIGN.md,
INTEGRATION.md).  Importable as `evstore_dlrm_amd` through the shim at the
repository root (the directory name carries a hyphen).
"""
import os as _os

# HIP folds a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise) and a queue runs its commands in
# order.  The resident exact-policy server (gpu_cache.GpuCache.serve_*) is a kernel that STAYS on its queue: a copy or a kernel
# of the caller's that is folded onto the same one waits until the server goes home idle (measured through the plug-in loop in
# a process with six streams: 299 us per request against 102 with eight queues).  A default, never an override; it counts only
# when the package is imported before the first call into the runtime.
if _os.environ.get("EVS_KEEP_HW_QUEUES", "0") != "1":   # (EVS_KEEP_HW_QUEUES=1: leave the runtime's own default alone)
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib, dlrm_ops, gpu_cache  # noqa: E402
from ._lib import EvsError, build  # noqa: E402
from .dlrm_ops import (EVTables, LazyPooled, apply_emb, apply_emb_interact, apply_emb_interact_mlp1, apply_emb_interact_multi, fused_supported,
                       interact_features, materialize)
from .gpu_cache import (FileTier, GpuAltKeyTier, GpuCache, lookup_batch_c1c2, lookup_batch_c1c2c3, lookup_interact_c1c2,
                        lookup_interact_c1c2c3, request_c1c2, request_c1c2c3)

__all__ = ["EvsError", "build", "EVTables", "LazyPooled", "apply_emb", "apply_emb_interact", "apply_emb_interact_multi", "apply_emb_interact_mlp1", "interact_features", "materialize", "fused_supported",
           "GpuCache", "FileTier", "GpuAltKeyTier", "request_c1c2", "request_c1c2c3", "lookup_batch_c1c2", "lookup_interact_c1c2",
           "lookup_batch_c1c2c3", "lookup_interact_c1c2c3"]
