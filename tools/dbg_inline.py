import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import evstore_dlrm_amd as E
from oracle import oracle as orc
from test_gpu_cache import _zipf_requests
n_rows = [3000, 40, 20000, 700, 5, 9000, 1500, 12, 26000, 300, 8000, 64, 2200, 17000, 3, 450, 5000, 90, 13000, 2, 7000, 30, 1000, 11000, 150, 4000]
tabs = orc.kaggle_tables(n_rows, 21)
dev = [torch.from_numpy(t).cuda() for t in tabs]
import itertools
nfail = 0
for seed, (cap_frac, batch) in itertools.product(range(int(sys.argv[1]) if len(sys.argv) > 1 else 12), ((0.02, 160), (0.02, 64), (0.1, 512))):
    if nfail >= 4: break
    cap = int(cap_frac * sum(n_rows))
    reqs = _zipf_requests(n_rows, 40 * batch, 5 + seed)
    x = torch.rand(batch, 36, device="cuda")
    c = E.GpuCache("evlfu", cap, 26, 36, 32, "python").set_batch_policy("setassoc"); c.set_backing(dev)
    r = torch.from_numpy(reqs).cuda()
    prev = {}
    for k in range(40):
        rq = reqs[k*batch:(k+1)*batch]
        hit, R = c.lookup_interact(r[k*batch:(k+1)*batch].contiguous(), x)
        d = c.batch_dump(); st = c.batch_stats()
        act = np.bincount(d[:, 0], minlength=27); h = np.array(st["hist"])
        if not np.array_equal(act, h):
            nfail += 1
            print(seed, cap_frac, batch, "batch", k, "diff idx", np.nonzero(act - h)[0], (act - h)[np.nonzero(act - h)[0]], "flush", st["n_flush"], "size", st["size"], len(d))
            cur = {(int(t), int(rw)): int(p) for p, t, rw in d}
            hitn = hit.cpu().numpy().astype(bool)
            # keys at prio lower than expected: requested this batch with agg
            agg = hitn.sum(1)
            for b in range(batch):
                for t in range(26):
                    key = (t+1, int(rq[b, t]))
                    if key in cur and cur[key] < agg[b]:
                        print("  key", key, "prio now", cur[key], "before", prev.get(key), "agg of its request", agg[b], "hit", hitn[b, t], "copies in batch", int((rq[:, t] == rq[b, t]).sum()), "aggs", agg[rq[:, t] == rq[b, t]], "hits", hitn[rq[:, t] == rq[b, t], t])
            break
        prev = {(int(t), int(rw)): int(p) for p, t, rw in d}
    else:
        pass
