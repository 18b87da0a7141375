#!/usr/bin/env python3
"""Developer stress of the resident dispatcher (InteractServer): random batch sizes posted in random bursts with random pauses
around the grid's idle time-out (so that posts land while the grid is leaving, gone, or coming back), stop() now and then,
other launches in between -- every R compared bit for bit with the launched kernel's.  Run under `timeout`; both front ends:
    python tools/fuzz_serve.py [seconds] [seed]            EVS_SERVE_PUBLISH=leader python tools/fuzz_serve.py ..."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import evstore_dlrm_amd as E

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
rs = np.random.RandomState(seed)
dev = torch.device("cuda")
t_end = time.time() + secs
n_cases = n_batches = 0
while time.time() < t_end:
    d = rnd.choice([16, 32, 36, 64])
    T = rnd.choice([1, 3, 8, 15, 16, 26, 27])
    ln = [int(v) for v in rs.randint(1, 5000, size=T)]
    ws = [torch.from_numpy(rs.uniform(-1, 1, size=(n, d)).astype(np.float32)).to(dev) for n in ln]
    ev = E.EVTables.from_fp32(ws)
    idle = rnd.choice([20, 60, 150, 400])
    srv = E.InteractServer(ev, n_blocks=rnd.choice([0, 0, 2, 3, 64, 300]), idle_us=idle)
    mode = "host-published" if srv.host_published else "leader"
    inflight = []          # (ticket, R, want)

    def drain():
        for tk, R, want in inflight:
            srv.wait(tk)
            assert torch.equal(R, want), ("R differs", mode, d, T, tuple(R.shape))
        inflight.clear()

    for _ in range(rnd.randint(20, 120)):
        B = rnd.choice([1, 1, 2, 15, 16, 17, 100, 257, 1000, 2048, 4097, 16384, rnd.randint(1, 20000)])
        x = torch.rand((B, d), device=dev)
        lS_i = torch.stack([torch.randint(0, n, (B,), device=dev, dtype=torch.int64) for n in ln])
        lS_o = torch.arange(B, device=dev, dtype=torch.int64).repeat(T, 1)
        torch.cuda.synchronize()          # (the inputs are complete; a device-wide synchronise also waits for an idle grid to leave)
        want = E.apply_emb_interact(x, lS_o, lS_i, ev)
        torch.cuda.synchronize()
        how = rnd.random()
        if how < 0.35:
            R = srv(x, lS_o, lS_i)
            assert torch.equal(R, want), ("R differs (call)", mode, d, T, B)
        else:
            tk, R = srv.post(x, lS_o, lS_i)
            inflight.append((tk, R, want))
            if len(inflight) >= rnd.choice([1, 2, 8, 40, 64, 100]):
                drain()
        n_batches += 1
        p = rnd.random()
        if p < 0.25:
            time.sleep(rnd.choice([0.2, 0.8, 1.0, 1.2, 3.0]) * idle * 1e-6)      # around the idle time-out
        elif p < 0.30:
            drain()
            srv.stop()
        elif p < 0.33:
            drain()
            srv.stop()
            torch.zeros(1000, device=dev).sum().item()                            # another launch needs the GPU
    drain()
    srv.stop()
    srv.close()
    E._lib.lib().evs_check_index_errors(None)
    n_cases += 1
print("serve fuzz ok: %d servers, %d batches in %.0f s (seed %d; last front end: %s)" % (n_cases, n_batches, secs, seed, mode))
