"""MI355X-native DLRM embedding lookup + EvLFU tiered cache + feature interaction.

Drop-in for the hot path of ucare-uchicago/ev-store-dlrm (see DESIGN.md,
INTEGRATION.md).  Importable as `evstore_dlrm_amd` through the shim at the
repository root (the directory name carries a hyphen).
"""
from . import _lib, dlrm_ops, gpu_cache
from ._lib import EvsError, build
from .dlrm_ops import (EVTables, LazyPooled, apply_emb, apply_emb_interact, apply_emb_interact_mlp1, apply_emb_interact_multi, fused_supported,
                       interact_features, materialize)
from .gpu_cache import (FileTier, GpuAltKeyTier, GpuCache, lookup_batch_c1c2, lookup_batch_c1c2c3, lookup_interact_c1c2,
                        lookup_interact_c1c2c3, request_c1c2, request_c1c2c3)

__all__ = ["EvsError", "build", "EVTables", "LazyPooled", "apply_emb", "apply_emb_interact", "apply_emb_interact_multi", "apply_emb_interact_mlp1", "interact_features", "materialize", "fused_supported",
           "GpuCache", "FileTier", "GpuAltKeyTier", "request_c1c2", "request_c1c2c3", "lookup_batch_c1c2", "lookup_interact_c1c2",
           "lookup_batch_c1c2c3", "lookup_interact_c1c2c3"]
