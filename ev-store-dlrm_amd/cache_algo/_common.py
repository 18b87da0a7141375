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
        # one request at a time, like the reference: ids, hit flags (and the rows when the caller wants them on
        # the host) live in pinned buffers the kernel reads / writes directly
        self._host_rows = torch.empty((1, self.n_tables), dtype=torch.int32).pin_memory()
        self._host_hit = torch.empty((1, self.n_tables), dtype=torch.uint8).pin_memory()
        self._host_out = torch.empty((1, self.n_tables, self.dim), dtype=torch.float32).pin_memory()
        self._dev_out = torch.empty((1, self.n_tables, self.dim), dtype=torch.float32, device=device)
        self._device = torch.device(device)
        self._bound = False

    def _bind(self):
        tabs = storage_manager.device_tables()
        if tabs is None:
            print("ERROR: the GPU cache reads misses from device-accessible storage: set "
                  "storage_manager.storage_type = EmbStorage.HBM or EmbStorage.PINNED before loading the tables")
            exit(-1)
        self.cache.set_backing(tabs)
        self._bound = True

    def request_rows(self, group_row_ids):
        """-> (list[bool] * T, (T, d) float32 numpy array on the host): the Cython module's return shape"""
        if self.cache is None:
            print("ERROR: call init(capacity) first")
            exit(-1)
        if not self._bound:
            self._bind()
        self._host_rows[0] = torch.as_tensor(group_row_ids, dtype=torch.int32)
        self.cache.request(self._host_rows, -1, out=self._host_out, hit=self._host_hit)
        torch.cuda.current_stream(self._device).synchronize()
        return [bool(v) for v in self._host_hit[0].tolist()], self._host_out[0].numpy().copy()

    def request(self, group_row_ids, use_gpu, approx_thres=-1):
        if self.cache is None:
            print("ERROR: call init(capacity) first")
            exit(-1)
        if not self._bound:
            self._bind()
        self._host_rows[0] = torch.as_tensor(group_row_ids, dtype=torch.int32)
        out = self._dev_out if use_gpu else self._host_out
        self.cache.request(self._host_rows, approx_thres, out=out, hit=self._host_hit)
        torch.cuda.current_stream(self._device).synchronize()
        arr_record_hit = [bool(v) for v in self._host_hit[0].tolist()]
        # 26 x Tensor(1, 36) with requires_grad, like the reference's torch.FloatTensor([val]) per table -- made as
        # ONE fresh (26, 1, 36) tensor and its 26 views (26 separate clones were most of this function's time)
        block = out[0].detach().clone().unsqueeze(1).requires_grad_(True)
        return arr_record_hit, list(block.unbind(0))
