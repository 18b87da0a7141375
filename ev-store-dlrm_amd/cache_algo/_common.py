"""Shared plumbing of the cache modules (module-global state, like the reference's).

Two engines behind the same surface, picked when the first request binds the storage manager's tables
(init(..., engine="auto")):
  host  the exact policy on the host (csrc/evs_hostcache.hip): a few microseconds per request, rows read from
        host-readable tables (EmbStorage.DUMMY / PINNED tensors, or the .bin files of FILEPY / MMAPFILEPY mapped
        read-only) -- the reference's batch-1 loop belongs here (docs/HISTORY.md 3.3);
  gpu   the GPU tier's exact kernel (csrc/evs_cache.hip) for tables that live in HBM only (EmbStorage.HBM), or when asked
        for (engine="gpu"): as a RESIDENT SERVER fed through a pinned-host mailbox when the rows are wanted on the device
        (use_gpu=True: ~25 us per request, rows never leave HBM), one launch + one synchronise per request otherwise.
Both give the same hit flags, rows and list order (tests/test_hostcache.py, tests/test_gpu_cache.py)."""
import numpy as np
import torch

from ..emb_storage import storage_manager


class _ModuleCache:
    def __init__(self, policy):
        self.policy = policy
        self.cache = None
        self.engine = None
        self._args = None
        self.n_tables, self.dim = storage_manager.N_EV_TABLE, storage_manager.EV_DIMENSION

    def init(self, capacity, variant="python", device="cuda", engine="auto"):
        assert engine in ("auto", "host", "gpu")
        self._args = (int(capacity), variant, device, engine)
        self.cache, self.engine, self._bound = None, None, False

    # ---- engines ------------------------------------------------------------------------------------------------------
    def _make_gpu(self, tabs):
        from .. import gpu_cache
        capacity, variant, device, _ = self._args
        self.cache = gpu_cache.GpuCache(self.policy, capacity, self.n_tables, self.dim, storage_manager.ev_precs, variant, device)
        # one request at a time, like the reference: ids, hit flags (and the rows when the caller wants them on
        # the host) live in pinned buffers the kernel reads / writes directly
        self._host_rows = torch.empty((1, self.n_tables), dtype=torch.int32).pin_memory()
        self._host_hit = torch.empty((1, self.n_tables), dtype=torch.uint8).pin_memory()
        self._host_out = torch.empty((1, self.n_tables, self.dim), dtype=torch.float32).pin_memory()
        self._dev_out = torch.empty((1, self.n_tables, self.dim), dtype=torch.float32, device=device)
        self._device = torch.device(device)
        self.cache.set_backing(tabs)
        self._serving = None      # approx_thres the resident server was armed with (None: not armed)
        self.engine = "gpu"

    def _make_host(self, tabs):
        from .. import host_cache
        capacity, variant, device, _ = self._args
        self.cache = host_cache.HostCache(self.policy, capacity, self.n_tables, self.dim, storage_manager.ev_precs, variant)
        self.cache.set_backing(tabs)
        self._rows = np.empty((1, self.n_tables), np.int32)
        self._hit = np.empty((1, self.n_tables), np.uint8)
        self._out = np.empty((1, self.n_tables, self.dim), np.float32)
        self._device = torch.device(device if device != "cuda" else "cuda:0")  # EvLFU_C1.py:132 hard-codes cuda:0
        self.engine = "host"

    def _bind(self):
        if self._args is None:
            print("ERROR: call init(capacity) first")
            exit(-1)
        want = self._args[3]
        host_tabs = storage_manager.host_tables()
        dev_tabs = storage_manager.device_tables()
        if want == "auto":
            want = "host" if host_tabs is not None else "gpu"
        if want == "host":
            if host_tabs is None:
                print("ERROR: the host cache engine reads misses from host-readable storage (EmbStorage.DUMMY / PINNED / "
                      "FILEPY / MMAPFILEPY); the tables are in HBM only -- use engine='gpu' or 'auto'")
                exit(-1)
            self._make_host(host_tabs)
        else:
            if dev_tabs is None:
                print("ERROR: the GPU cache reads misses from device-accessible storage: set "
                      "storage_manager.storage_type = EmbStorage.HBM or EmbStorage.PINNED before loading the tables")
                exit(-1)
            self._make_gpu(dev_tabs)
        self._bound = True

    def stats(self):
        """the cache's counters; binds the engine first when nothing has been requested yet (init() only records its
        arguments: the engine is chosen when the storage manager's tables are known)"""
        if self.cache is None:
            self._bind()
        return self.cache.stats()

    # ---- requests -----------------------------------------------------------------------------------------------------
    def _run(self, group_row_ids, approx_thres, want_device_rows):
        """-> (hit flags as a list of bool, rows as a (T, d) float32 tensor on the host, or on the device when the GPU
        engine can leave them there)"""
        if not getattr(self, "_bound", False):
            self._bind()
        if self.engine == "host":
            self._rows[0] = group_row_ids
            self.cache.request(self._rows, approx_thres, out=self._out, hit=self._hit)
            return self._hit[0].astype(bool).tolist(), torch.from_numpy(self._out[0])
        if want_device_rows and self.n_tables <= 28:
            # the resident server (round 5; gpu_cache.GpuCache.serve_*): the ids go out and the hit flags come back through a
            # mailbox in pinned host memory, the rows stay in a ring in HBM -- no launch, no copy, no synchronise per request
            if self._serving != approx_thres:
                self.cache.serve_start(approx_thres, n_slots=4, idle_us=200)
                self._serving = approx_thres
            hit, rows = self.cache.serve_request(group_row_ids)
            self._ring_rows = True     # (request() says when its copy of the slot has been enqueued: serve_consumed)
            return hit.astype(bool).tolist(), rows
        self._host_rows[0] = torch.as_tensor(group_row_ids, dtype=torch.int32)
        out = self._dev_out if want_device_rows else self._host_out
        self.cache.request(self._host_rows, approx_thres, out=out, hit=self._host_hit)
        torch.cuda.current_stream(self._device).synchronize()
        return [bool(v) for v in self._host_hit[0].tolist()], out[0]

    def request_rows(self, group_row_ids):
        """-> (list[bool] * T, (T, d) float32 numpy array on the host): the Cython module's return shape"""
        hit, rows = self._run(group_row_ids, -1, False)
        return hit, rows.numpy().copy()

    def request_from_index_rows(self, lS_i, use_gpu, approx_thres=-1):
        """apply_emb_evstore's whole body for the host engine in ONE extension call: element 0 of each table's index row
        (dlrm_s_pytorch_C1.py:236-239), the exact policy, the 26 x Tensor(1, 36) list.  -> (hit flags, ly, perfect) or None
        when this request has to take the generic path."""
        if not getattr(self, "_bound", False):
            self._bind()
        if not torch.is_tensor(lS_i) or lS_i.dtype not in (torch.int64, torch.int32) \
                or lS_i.dim() < 1 or lS_i.shape[0] != self.n_tables or lS_i.numel() < self.n_tables \
                or (lS_i.is_cuda and (lS_i.dtype != torch.int64 or not use_gpu)):
            return None
        from .. import _ext
        X = _ext.ext()
        if X is None:
            return None
        if self.engine != "host":
            # the GPU engine's resident server (see _run): the same one-call body -- ids through pinned staging, the request
            # through the mailbox, the 26 tensors over one copy of the answer's ring slot
            if not use_gpu or self.n_tables > 28 or not hasattr(X, "serve_request_list"):
                return None
            if self._serving != approx_thres:
                self.cache.serve_start(approx_thres, n_slots=4, idle_us=200)
                self._serving = approx_thres
            return X.serve_request_list(self.cache._h.value, lS_i, self.cache.serve_ring, self.n_tables, self.dim)
        return X.hostcache_request_list(self.cache._h.value, lS_i, self.n_tables, self.dim, int(approx_thres), bool(use_gpu),
                                        self._device.index or 0)

    def request(self, group_row_ids, use_gpu, approx_thres=-1):
        arr_record_hit, rows = self._run(group_row_ids, approx_thres, use_gpu)
        # 26 x Tensor(1, 36) with requires_grad, like the reference's torch.FloatTensor([val]) per table -- made as
        # ONE fresh (26, 1, 36) tensor and its 26 views (26 separate clones were most of this function's time)
        if use_gpu and not rows.is_cuda:
            block = rows.to(self._device, copy=True)   # ONE host-to-device copy instead of 26 (EvLFU_C1.py:157-161)
        else:
            block = rows.detach().clone()
            if getattr(self, "_ring_rows", False):   # the clone of a ring slot is only enqueued: the slot is reused after it has run
                self.cache.serve_consumed()
                self._ring_rows = False
        from .. import _ext
        X = _ext.ext()
        if X is not None:
            return arr_record_hit, X.slices(block.unsqueeze(1), True)   # 26 leaf tensors over one block
        block = block.unsqueeze(1).requires_grad_(True)
        return arr_record_hit, list(block.unbind(0))
