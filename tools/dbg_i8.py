import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import evstore_dlrm_amd as E
from oracle import oracle as orc
from bench import KAGGLE_LN
d, B, codec = 36, 301, 8
rs = np.random.RandomState(78)
ln = [min(n, 333) | 1 for n in KAGGLE_LN]
raws = [orc.encode_table(rs.uniform(-1, 1, size=(n, d)).astype(np.float32), codec) for n in ln]
ev = E.EVTables([torch.from_numpy(r).cuda() for r in raws], d, codec)
idx_np = np.stack([rs.randint(0, n, size=B) for n in ln]).astype(np.int64)
idx = torch.from_numpy(idx_np).cuda()
off = torch.arange(B, device="cuda").repeat(26, 1)
x_np = rs.uniform(-1, 1, size=(B, d)).astype(np.float32)
x = torch.from_numpy(x_np).cuda()
a = E.apply_emb_interact(x, off, idx, ev, check_indices=True)
b = E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True, check_indices=True)
a2 = E.apply_emb_interact(x, off, idx, ev, check_indices=True)
print("a==a2", torch.equal(a, a2))
diff = (a != b)
print("rows differing", diff.any(1).nonzero().flatten().tolist()[:40], "cols", diff.any(0).nonzero().flatten().tolist()[:40])
print("max abs diff", (a - b).abs().max().item())
ly = orc.apply_emb([np.arange(B, dtype=np.int64)] * 26, list(idx_np), raws, None, codec, d)
want = orc.interact_features(x_np, ly)
for nm, t in (("a", a), ("b", b)):
    err = np.abs(t.cpu().numpy() - want); tol = 2e-6 + 1e-5 * np.abs(want)
    print(nm, "max err/tol", (err / tol).max(), "argmax", np.unravel_index((err / tol).argmax(), err.shape))
