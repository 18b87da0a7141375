"""The reference's inference harness around the hot path -- the parts that define what its
"per-request latency" and "p50 batch latency" mean:

  dlrm_wrap(X, lS_o, lS_i, use_gpu, device)      dlrm_s_pytorch.py:131-147  per-batch H2D of X, lS_o, lS_i, then forward
  inference(...)                                 dlrm_s_pytorch_C1.py:921-1118  one wall-clock stamp at the TOP of every
                                                 request (:965) plus one after the last (:1052); a request's latency is
                                                 the difference of consecutive stamps, so it includes the loader and
                                                 whatever the loop does with Z
  warm-up                                        dlrm_s_pytorch_C1.py:2224-2242  one full replay of the same workload
  calculate_and_write_cdf(dir, algo, stamps)     dlrm_s_pytorch_C1.py:299-326  sorted latencies thinned to ~1000 points,
                                                 CSV columns y, latency_ms

Host-side plumbing only: PyTorch moves the bytes, the forward it wraps is this package's HIP path.
"""
import os
import time

import torch


def dlrm_wrap(forward, X, lS_o, lS_i, use_gpu, device, non_blocking=False):
    """dlrm_s_pytorch.py:131-147: move the batch to the device (lists tensor by tensor, stacked tensors whole), then
    forward(X, lS_o, lS_i).  non_blocking=True is what a pinned-memory loader gets (dlrm_data_pytorch pin_memory)."""
    if use_gpu:
        lS_i = [S_i.to(device, non_blocking=non_blocking) for S_i in lS_i] if isinstance(lS_i, list) \
            else lS_i.to(device, non_blocking=non_blocking)
        lS_o = [S_o.to(device, non_blocking=non_blocking) for S_o in lS_o] if isinstance(lS_o, list) \
            else lS_o.to(device, non_blocking=non_blocking)
        X = X.to(device, non_blocking=non_blocking)
    return forward(X, lS_o, lS_i)


def inference(test_ld, forward, use_gpu=True, device="cuda", consume=None, non_blocking=False):
    """The timing loop of dlrm_s_pytorch_C1.py:inference(): stamps at the top of every request and one after the last.
    test_ld yields (X, lS_o, lS_i) host batches; consume(Z) stands for what the loop does with the result (the
    reference copies Z to the host, :1025, unless --ev-lookup-only).  -> arr_time_start (len = requests + 1)."""
    arr_time_start = []
    for X, lS_o, lS_i in test_ld:
        arr_time_start.append(time.time())
        Z = dlrm_wrap(forward, X, lS_o, lS_i, use_gpu, device, non_blocking)
        if consume is not None:
            consume(Z)
    arr_time_start.append(time.time())   # completion of the last request (:1052)
    return arr_time_start


def latencies(arr_time_start):
    """Per-request latencies in seconds as calculate_and_write_cdf derives them (the reference's loop bound drops the
    last request: range(0, len - 2), dlrm_s_pytorch_C1.py:304)."""
    return [arr_time_start[i + 1] - arr_time_start[i] for i in range(0, len(arr_time_start) - 2)]


def calculate_and_write_cdf(cdf_output_dir, cache_algo, arr_time_start, n_points=1000):
    """dlrm_s_pytorch_C1.py:299-326 without the plot subprocess: <dir>/<cache_algo>-cdf.csv with columns y, latency_ms
    (sorted latencies, every int(n/1000)-th kept, y = rank / points).  Returns the path."""
    os.makedirs(cdf_output_dir, exist_ok=True)
    arr_latency = sorted(latencies(arr_time_start))
    n_rows = len(arr_latency)
    step = int(n_rows / n_points)
    if step >= 1:   # (the reference divides by zero below 1000 requests; here short runs keep every point)
        arr_latency = arr_latency[0::step]
    output = os.path.join(cdf_output_dir, cache_algo + "-cdf.csv")
    import pandas as pd   # the reference writes through pandas; same frame, same to_csv call
    df = pd.DataFrame(arr_latency, columns=["latency_ms"])
    df["latency_ms"] = df["latency_ms"] * 1000
    df["y"] = df.index.values
    df["y"] = df["y"] + 1
    df["y"] = df["y"] / df.shape[0]
    df = df[["y", "latency_ms"]]
    df.to_csv(output, sep=",", index=False)
    print("CDF Latency data points is written to: " + output)
    return output


def percentile_ms(arr_time_start, q):
    import numpy as np
    lat = latencies(arr_time_start)
    return float(np.percentile(lat, q)) * 1e3 if lat else float("nan")


class PinnedBatches:
    """A loader stand-in: n host batches in PINNED memory (what DataLoader(pin_memory=True) hands the loop),
    cycled for `count` requests."""

    def __init__(self, batches, count):
        self.batches = [tuple(t.pin_memory() if torch.is_tensor(t) else [u.pin_memory() for u in t] for t in b)
                        for b in batches]
        self.count = count

    def __len__(self):
        return self.count

    def __iter__(self):
        for i in range(self.count):
            yield self.batches[i % len(self.batches)]


class PackedPinnedBatches:
    """A loader stand-in for the THROUGHPUT form of the loop: every batch is ONE pinned block -- X (B, 13) fp32 | lS_o (T, B) |
    lS_i (T, B) -- so that it crosses the bus as ONE copy command instead of the three of dlrm_wrap
    (dlrm_s_pytorch.py:131-147); the tensors a batch yields are views of its block.  index_dtype=torch.int32 is the opt-in
    narrow wire format: offsets and indices travel as 4 bytes (Criteo's row ids fit: the largest Kaggle table has 10.1 M
    rows) and are widened to the int64 the kernels -- and the reference -- read by a device-side copy: 468 -> 260 bytes per
    sample at T = 26."""

    def __init__(self, batches, count, index_dtype=torch.int64):
        assert index_dtype in (torch.int64, torch.int32)
        self.index_dtype, self.count = index_dtype, count
        self.blocks, self.layout = [], None
        isz = 8 if index_dtype == torch.int64 else 4
        for X, lS_o, lS_i in batches:
            X, lS_o, lS_i = X.contiguous(), lS_o.contiguous(), lS_i.contiguous()
            if index_dtype == torch.int32 and lS_i.numel():
                assert int(lS_i.max()) < 2 ** 31 and int(lS_o.max()) < 2 ** 31 and int(lS_i.min()) >= -2 ** 31, \
                    "the int32 wire format needs row ids and offsets below 2^31"
            nx = (X.numel() * 4 + 15) // 16 * 16
            no = lS_o.numel() * isz
            lay = (tuple(X.shape), tuple(lS_o.shape), tuple(lS_i.shape), nx, no)
            assert self.layout in (None, lay), "every batch must have the same shape"
            self.layout = lay
            blk = torch.empty(nx + no + lS_i.numel() * isz, dtype=torch.uint8).pin_memory()
            blk[:X.numel() * 4].view(torch.float32).view(X.shape).copy_(X)
            blk[nx:nx + no].view(index_dtype).view(lS_o.shape).copy_(lS_o)
            blk[nx + no:].view(index_dtype).view(lS_i.shape).copy_(lS_i)
            self.blocks.append(blk)
        self.nbytes = int(self.blocks[0].numel())

    def views(self, blk):
        """(X, lS_o, lS_i) of a block (host or device), no copy"""
        xs, os_, is_, nx, no = self.layout
        n_x = xs[0] * xs[1] * 4
        return (blk[:n_x].view(torch.float32).view(xs), blk[nx:nx + no].view(self.index_dtype).view(os_),
                blk[nx + no:].view(self.index_dtype).view(is_))

    def __len__(self):
        return self.count


class PackedRaggedPinnedBatches:
    """PackedPinnedBatches for the reference's RANDOM-data loader (RandomDataset + collate_wrapper_random_offset,
    dlrm_data_pytorch.py:678-797): multi-hot batches -- X (B, m_den) fp32, lS_o (T, B) int64 bag starts, lS_i a LIST of T int64
    tensors of different lengths.  Every batch is ONE pinned block -- X | lS_o | the T index tensors back to back (each
    starting on 16 bytes) -- one copy command; the tensors a batch yields are views of it (lS_i: a list of T views).  Batches
    may differ in their numbers of indices: the prefetcher's slots take the largest."""
    index_dtype = torch.int64
    ragged = True

    def __init__(self, batches, count):
        self.count, self.blocks, self.layouts = count, [], []
        for X, lS_o, lS_i in batches:
            X = X.contiguous()
            lS_o = (torch.stack(list(lS_o)) if not torch.is_tensor(lS_o) else lS_o).contiguous()
            nx = (X.numel() * 4 + 15) // 16 * 16
            no = lS_o.numel() * 8
            offs, at = [], nx + no
            for t in lS_i:
                offs.append((at, int(t.numel())))
                at += (int(t.numel()) * 8 + 15) // 16 * 16
            blk = torch.empty(max(at, 16), dtype=torch.uint8).pin_memory()
            blk[:X.numel() * 4].view(torch.float32).view(X.shape).copy_(X)
            blk[nx:nx + no].view(torch.int64).view(lS_o.shape).copy_(lS_o)
            for (o, n), t in zip(offs, lS_i):
                if n:
                    blk[o:o + 8 * n].view(torch.int64).copy_(t.to(torch.int64))
            self.blocks.append(blk)
            self.layouts.append((tuple(X.shape), tuple(lS_o.shape), nx, no, offs))
        self.nbytes = max(int(b.numel()) for b in self.blocks)

    def views_of(self, k, blk):
        """(X, lS_o, [lS_i_0 .. lS_i_{T-1}]) of batch k's block (host or device), no copy"""
        xs, os_, nx, no, offs = self.layouts[k]
        return (blk[:xs[0] * xs[1] * 4].view(torch.float32).view(xs), blk[nx:nx + no].view(torch.int64).view(os_),
                [blk[o:o + 8 * n].view(torch.int64) for o, n in offs])

    def __len__(self):
        return self.count


def collate_criteo_offset(x_int, x_cat, X=None, lS_o=None, lS_i=None, write_offsets=True, max_ind_range=-1):
    """collate_wrapper_criteo_offset (dlrm_data_pytorch.py:397-410) on the DEVICE: the raw batch as CriteoDataset.__getitem__
    yields it -- x_int (B, n_dense) int32 counts, x_cat (B, T) int32 ids, both on the GPU -- to (X, lS_o, lS_i) as the loader
    hands them to dlrm_wrap: X = log(x_int + 1) fp32, lS_o = arange(B) per table and lS_i = x_cat transposed, (T, B) int64.
    One launch (evs_collate_criteo_offset); buffers are re-used when given (write_offsets=False: lS_o already holds arange).
    x_int / x_cat may be column views of one record array (rows contiguous, any row stride): the Terabyte binary loader's
    (B, 40) blocks (script/data_loader_terabyte.py:226-236) -- see collate_criteo_records; max_ind_range > 0: ids modulo it
    (_transform_features, :71-72)."""
    from . import _lib
    from .dlrm_ops import _stream_ptr
    assert x_int.is_cuda and x_cat.is_cuda and x_int.dtype == torch.int32 and x_cat.dtype == torch.int32
    if x_int.dim() != 2 or (x_int.shape[1] > 1 and x_int.stride(1) != 1):
        x_int = x_int.contiguous()
    if x_cat.dim() != 2 or (x_cat.shape[1] > 1 and x_cat.stride(1) != 1):
        x_cat = x_cat.contiguous()
    B, nd, T = int(x_cat.shape[0]), int(x_int.shape[1]), int(x_cat.shape[1])
    dev = x_cat.device
    X = torch.empty((B, nd), dtype=torch.float32, device=dev) if X is None else X
    lS_i = torch.empty((T, B), dtype=torch.int64, device=dev) if lS_i is None else lS_i
    if lS_o is None:
        lS_o, write_offsets = torch.empty((T, B), dtype=torch.int64, device=dev), True
    assert X.is_contiguous() and lS_i.is_contiguous() and lS_o.is_contiguous() and X.shape == (B, nd) and lS_i.shape == (T, B) == lS_o.shape
    si = int(x_int.stride(0)) if B > 1 else max(nd, 1)
    sc = int(x_cat.stride(0)) if B > 1 else T
    _lib.check(_lib.lib().evs_collate_criteo_offset(B, nd, T, x_int.data_ptr(), max(si, nd), x_cat.data_ptr(), max(sc, T), int(max_ind_range),
                                                    X.data_ptr(), lS_o.data_ptr() if write_offsets else None, lS_i.data_ptr(), _stream_ptr(dev)))
    return X, lS_o, lS_i


def collate_criteo_records(rec, **kw):
    """the Terabyte binary loader's batch -- a (B, 40) int32 block of the file: label, 13 counts, 26 ids per record
    (CriteoBinDataset.__getitem__, script/data_loader_terabyte.py:226-236) -- collated on the device; -> X, lS_o, lS_i"""
    assert rec.dim() == 2 and rec.shape[1] == 40 and rec.dtype == torch.int32
    return collate_criteo_offset(rec[:, 1:14], rec[:, 14:], **kw)


class RawCriteoPinnedBatches:
    """A loader stand-in that keeps the batches RAW, as the dataset holds them: every batch ONE pinned block -- x_int (B, n_dense)
    int32 | x_cat (B, T) int32 -- 156 bytes per sample at (13, 26) instead of the 468 of the collated (X, lS_o, lS_i); the
    Prefetcher collates each batch on the device behind its copy (collate_criteo_offset).  batches: (x_int, x_cat) integer
    arrays / tensors of one shape."""
    raw = True
    index_dtype = torch.int64

    def __init__(self, batches, count, max_ind_range=-1):
        self.count, self.blocks, self.layout, self.max_ind_range = count, [], None, max_ind_range
        for x_int, x_cat in batches:
            x_int = torch.as_tensor(x_int).to(torch.int32).contiguous()
            x_cat = torch.as_tensor(x_cat).to(torch.int32).contiguous()
            n1 = (x_int.numel() * 4 + 15) // 16 * 16
            lay = (tuple(x_int.shape), tuple(x_cat.shape), n1)
            assert self.layout in (None, lay) and x_int.shape[0] == x_cat.shape[0], "every batch must have the same shape"
            self.layout = lay
            blk = torch.empty(n1 + x_cat.numel() * 4, dtype=torch.uint8).pin_memory()
            blk[:x_int.numel() * 4].view(torch.int32).view(x_int.shape).copy_(x_int)
            blk[n1:].view(torch.int32).view(x_cat.shape).copy_(x_cat)
            self.blocks.append(blk)
        self.nbytes = int(self.blocks[0].numel())

    def views(self, blk):
        """(x_int, x_cat) of a block (host or device), no copy"""
        s1, s2, n1 = self.layout
        return blk[:s1[0] * s1[1] * 4].view(torch.int32).view(s1), blk[n1:].view(torch.int32).view(s2)

    def __len__(self):
        return self.count


class RawCriteoRecordBatches(RawCriteoPinnedBatches):
    """The same for the Terabyte binary loader: every batch is the (B, 40) int32 block CriteoBinDataset reads from its file
    (script/data_loader_terabyte.py:226-236: label, 13 counts, 26 ids per record; 160 bytes per sample), pinned as it is;
    the prefetcher collates it on the device (collate_criteo_records, ids modulo max_ind_range when > 0)."""

    def __init__(self, records, count, max_ind_range=-1):
        self.count, self.blocks, self.max_ind_range = count, [], max_ind_range
        for rec in records:
            rec = torch.as_tensor(rec).to(torch.int32).contiguous()
            assert rec.dim() == 2 and rec.shape[1] == 40, "records are (B, 40) int32"
            self.layout = ((int(rec.shape[0]), 13), (int(rec.shape[0]), 26), 0)
            blk = torch.empty(rec.numel() * 4, dtype=torch.uint8).pin_memory()
            blk.view(torch.int32).view(rec.shape).copy_(rec)
            assert not self.blocks or blk.numel() == self.blocks[0].numel(), "every batch must have the same shape"
            self.blocks.append(blk)
        self.nbytes = int(self.blocks[0].numel())

    def views(self, blk):
        rec = blk.view(torch.int32).view(self.layout[0][0], 40)
        return rec[:, 1:14], rec[:, 14:]


class Prefetcher:
    """H2D staging for PackedPinnedBatches: ONE copy command per batch into a rotating device slot; the tensors a batch
    yields are views of its slot (int32 wire batches arrive widened to int64), valid until the slot comes round again.
      copy_stream=False: the copy is queued on the caller's stream in front of the batch's launches -- no
        synchronisation, no per-request stamps: the bus stays busy back to back (7.7 MB per 16 384-batch in 0.144 ms, the
        20 us launch behind it: 0.165-0.172 ms per batch against 0.193 for the reference loop's three copies + synchronise);
      copy_stream=True (the default where the device has stream wait-value operations: 0.160 ms per batch = 47.8 GB/s):
        batch i + 1 crosses on a copy stream under batch i's launch.  The two hand-overs per batch (copied
        -> compute may start; consumed -> the slot may be overwritten) are SIGNAL WORDS written and waited for by the
        command processors in stream order (evs_stream_write_value / evs_stream_wait_value: csrc/evs_p2p.hip) -- an event
        wait between two streams costs more (0.175-0.18 ms per batch against 0.160, and 0.3-0.5 ms when the issuing core
        has been idle) -- with events as the fall-back where the device has no such operations (signals=False forces it).
    for X, lS_o, lS_i in Prefetcher(ld, device): forward(...)"""

    def __init__(self, ld, device, depth=2, copy_stream=None, signals=True):
        self.ld, self.device, self.depth = ld, torch.device(device), depth
        auto = copy_stream is None     # default: the copy stream when its hand-overs can be signal words, else one stream
        if auto:
            copy_stream = bool(signals)
        self.cs = torch.cuda.Stream(device=self.device) if copy_stream else None
        self.slots = [torch.empty(ld.nbytes, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.copied = [torch.cuda.Event() for _ in range(depth)]
        self.used = [torch.cuda.Event() for _ in range(depth)]
        self.sig = None          # per slot: (copied word, used word), counting the slot's uses
        self.uses = [0] * depth
        self.recorded = [False] * depth   # event hand-overs: the slot's `used` event has been recorded at least once
        self._chk = lambda rc: rc        # (replaced by _lib.check where the signal words are in use)
        if copy_stream and signals:
            from . import _lib
            import ctypes as C
            L = _lib.lib()
            sig = []
            try:
                with torch.cuda.device(self.device):
                    for _ in range(2 * depth):
                        p = C.c_void_p()
                        _lib.check(L.evs_signal_alloc(C.byref(p)))
                        sig.append(p.value)
                self.sig = [(sig[2 * i], sig[2 * i + 1]) for i in range(depth)]
                self._L = L
                self._chk = _lib.check   # a failed hipStreamWaitValue32 / WriteValue32 raises: a dropped hand-over is a data race
            except Exception:    # no stream wait-value operations on this device: events (or, by default, one stream)
                for p in sig:
                    L.evs_signal_free(C.c_void_p(p))
                self.sig = None
                if auto:
                    self.cs = None
        self.cooked = None       # raw batches: per slot the collated (X, lS_o, lS_i) the device-side collate writes
        if getattr(ld, "raw", False):
            (B_, nd_), (_, T_), _ = ld.layout
            ar = torch.arange(B_, dtype=torch.int64, device=self.device).repeat(T_, 1).contiguous()
            self.cooked = [(torch.empty((B_, nd_), dtype=torch.float32, device=self.device), ar.clone(),
                            torch.empty((T_, B_), dtype=torch.int64, device=self.device)) for _ in range(depth)]
        self.wide = None
        if ld.index_dtype == torch.int32:
            _, os_, is_, _, _ = ld.layout
            self.wide = [(torch.empty(os_, dtype=torch.int64, device=self.device), torch.empty(is_, dtype=torch.int64, device=self.device))
                         for _ in range(depth)]

    def __del__(self):
        if getattr(self, "sig", None):
            try:
                torch.cuda.synchronize(self.device)
                for a, b in self.sig:
                    self._L.evs_signal_free(a)
                    self._L.evs_signal_free(b)
            except Exception:
                pass
            self.sig = None

    def _issue(self, i):
        sl = i % self.depth
        with torch.cuda.stream(self.cs):
            # the slot's previous use has been consumed
            if self.sig:
                self._chk(self._L.evs_stream_wait_value(self.cs.cuda_stream, self.sig[sl][1], self.uses[sl] & 0xffffffff))
            elif self.recorded[sl]:    # (whichever pass used the slot last: a second pass over this object starts at i = 0 again)
                self.cs.wait_event(self.used[sl])
            src = self.ld.blocks[i % len(self.ld.blocks)]
            self.slots[sl][:src.numel()].copy_(src, non_blocking=True)
            self.uses[sl] += 1
            if self.sig:
                self._chk(self._L.evs_stream_write_value(self.cs.cuda_stream, self.sig[sl][0], self.uses[sl] & 0xffffffff))
            else:
                self.copied[sl].record(self.cs)

    def _views(self, sl, i=0):
        if getattr(self.ld, "ragged", False):
            return self.ld.views_of(i % len(self.ld.blocks), self.slots[sl])
        if self.cooked is not None:   # (lS_o was written once, when the slot was made)
            xi, xc = self.ld.views(self.slots[sl])
            X, lo, li = self.cooked[sl]
            collate_criteo_offset(xi, xc, X=X, lS_o=lo, lS_i=li, write_offsets=False, max_ind_range=getattr(self.ld, "max_ind_range", -1))
            return X, lo, li
        X, lo, li = self.ld.views(self.slots[sl])
        if self.wide is not None:
            wo, wi = self.wide[sl]
            wo.copy_(lo)
            wi.copy_(li)
            lo, li = wo, wi
        return X, lo, li

    def __iter__(self):
        n = len(self.ld)
        if self.cs is None:   # one stream: copy, then the caller's launches, in stream order
            for i in range(n):
                sl = i % self.depth
                src = self.ld.blocks[i % len(self.ld.blocks)]
                self.slots[sl][:src.numel()].copy_(src, non_blocking=True)
                yield self._views(sl, i)
            return
        main = torch.cuda.current_stream(self.device)
        self._base = list(self.uses)
        try:
            for i in range(min(self.depth - 1, n)):
                self._issue(i)
            for i in range(n):
                if i + self.depth - 1 < n:
                    self._issue(i + self.depth - 1)      # the next batch starts crossing now
                sl = i % self.depth
                if self.sig:   # (the value the copy of batch i wrote: its slot's use count)
                    self._chk(self._L.evs_stream_wait_value(main.cuda_stream, self.sig[sl][0], self._use_of(i) & 0xffffffff))
                else:
                    main.wait_event(self.copied[sl])
                yield self._views(sl, i)
                if self.sig:                              # (behind whatever the caller queued on its stream for this batch)
                    self._chk(self._L.evs_stream_write_value(main.cuda_stream, self.sig[sl][1], self._use_of(i) & 0xffffffff))
                else:
                    self.used[sl].record(main)
                    self.recorded[sl] = True
        finally:
            # a pass left early: every slot issued counts as consumed behind what the caller has queued so far (nobody waits for a
            # hand-over that never comes, and the next pass's copies wait for the launches of the batch the caller broke out of)
            for sl in range(self.depth):
                if self.sig:
                    self._chk(self._L.evs_stream_write_value(main.cuda_stream, self.sig[sl][1], self.uses[sl] & 0xffffffff))
                else:
                    self.used[sl].record(main)
                    self.recorded[sl] = True

    def _use_of(self, i):
        """the use count of slot i % depth when batch i of the current pass was issued"""
        return self._base[i % self.depth] + i // self.depth + 1
