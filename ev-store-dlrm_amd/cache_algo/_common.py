"""Shared plumbing of the three cache modules (module-global state, like the reference's)."""
import torch

from .. import gpu_cache
from ..emb_storage import storage_manager


class _ModuleCache:
    def __init__(self, policy):
        self.policy = policy
        self.cache = None
        self.rows = None
        self.n_tables, self.dim = storage_manager.N_EV_TABLE, storage_manager.EV_DIMENSION

    def init(self, capacity, variant="python", device="cuda"):
        self.cache = gpu_cache.GpuCache(self.policy, capacity, self.n_tables, self.dim,
                                        storage_manager.ev_precs, variant, device)
        self.rows = torch.empty((1, self.n_tables), dtype=torch.int32, device=device)
        self._host_rows = torch.empty((1, self.n_tables), dtype=torch.int32).pin_memory()
        self._bound = False

    def _bind(self):
        tabs = storage_manager.device_tables()
        if tabs is None:
            print("ERROR: the GPU cache reads misses from device-accessible storage: set "
                  "storage_manager.storage_type = EmbStorage.HBM or EmbStorage.PINNED before loading the tables")
            exit(-1)
        self.cache.set_backing(tabs)
        self._bound = True

    def request(self, group_row_ids, use_gpu, approx_thres=-1):
        if self.cache is None:
            print("ERROR: call init(capacity) first")
            exit(-1)
        if not self._bound:
            self._bind()
        self._host_rows[0] = torch.as_tensor(group_row_ids, dtype=torch.int32)
        self.rows.copy_(self._host_rows, non_blocking=True)
        hit, out = self.cache.request(self.rows, approx_thres)
        arr_record_hit = [bool(v) for v in hit[0].tolist()]
        vals = out[0] if use_gpu else out[0].cpu()
        arr_emb_weights = []
        for k in range(self.n_tables):
            t = vals[k:k + 1].detach().clone()   # Tensor(1, 36), as torch.FloatTensor([val])
            t.requires_grad = True
            arr_emb_weights.append(t)
        return arr_record_hit, arr_emb_weights
