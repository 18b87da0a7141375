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
