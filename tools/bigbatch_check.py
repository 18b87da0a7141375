"""Developer check: the fused launch at a very large batch (2^20 samples) against the two-call path (values)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, evstore_dlrm_amd as E
ev = bench.make_tables(bench.KAGGLE_LN, 36)
B = 1 << 20
g = torch.Generator(device="cuda").manual_seed(3)
idx = torch.stack([torch.randint(0, n, (B,), device="cuda", generator=g) for n in bench.KAGGLE_LN])
off = torch.arange(B, device="cuda").repeat(26, 1)
x = torch.randn(B, 36, device="cuda")
a = E.apply_emb_interact(x, off, idx, ev, check_indices=True)
b = E.apply_emb_interact(x, off, idx, ev, one_index_per_bag=True)
assert torch.equal(a, b) and torch.equal(a[:, :36], x)
for k in (0, 20):
    f = k + 1; col = 36 + f * (f - 1) // 2
    want = (ev.fp32_view(k)[idx[k]].double() * x.double()).sum(1)
    torch.testing.assert_close(a[:, col].double(), want, rtol=1e-5, atol=1e-6)
# the last sample, recomputed whole
T = torch.stack([x[-1]] + [ev.fp32_view(k)[idx[k, -1]] for k in range(26)]).double()
Z = T @ T.t()
li, lj = torch.tril_indices(27, 27, offset=-1)
torch.testing.assert_close(a[-1, 36:].double().cpu(), Z[li, lj].cpu(), rtol=1e-5, atol=1e-6)
print("B=2^20 ok")
