"""MI355X-native DLRM embedding lookup + EvLFU tiered cache + feature interaction.

Drop-in for the hot path of ucare-uchicago/ev-store-dlrm (see DESIGN.md,
INTEGRATION.md).  Importable as `evstore_dlrm_amd` through the shim at the
repository root (the directory name carries a hyphen).
"""
from . import _lib
from ._lib import EvsError, build
from .dlrm_ops import EVTables, apply_emb, apply_emb_interact, interact_features
from .gpu_cache import GpuCache

__all__ = ["EvsError", "build", "GpuCache", "EVTables", "apply_emb", "apply_emb_interact", "interact_features"]
