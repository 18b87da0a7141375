"""TEST INFRASTRUCTURE -- the reference's CPU path restated with the same torch CPU ops.

Same per-table loop and the same ATen operators as DLRM_Net.apply_emb
(dlrm_s_pytorch.py:407-461: nn.EmbeddingBag(mode="sum") per table) and
DLRM_Net.interact_features (dlrm_s_pytorch.py:483-516: cat, bmm, index gather, cat).
Used only as bench.py's cpu_baseline ("port") and by tests; checked against the
golden vectors in tests/test_oracle_golden.py::test_cpu_port_matches_golden.
"""
import torch
import torch.nn as nn


class CpuHotPath:
    def __init__(self, tables, arch_interaction_itself=False):
        self.emb_l = nn.ModuleList()
        for w in tables:
            n, m = w.shape
            e = nn.EmbeddingBag(n, m, mode="sum", sparse=True)  # dlrm_s_pytorch.py:276
            e.weight.data = w
            self.emb_l.append(e)
        self.v_W_l = [None] * len(tables)
        self.itself = arch_interaction_itself

    def apply_emb(self, lS_o, lS_i):
        ly = []
        for k, sparse_index_group_batch in enumerate(lS_i):
            sparse_offset_group_batch = lS_o[k]
            w = self.v_W_l[k]
            psw = w.gather(0, sparse_index_group_batch) if w is not None else None
            ly.append(self.emb_l[k](sparse_index_group_batch, sparse_offset_group_batch, per_sample_weights=psw))
        return ly

    def interact_features(self, x, ly):
        (batch_size, d) = x.shape
        T = torch.cat([x] + ly, dim=1).view((batch_size, -1, d))
        Z = torch.bmm(T, torch.transpose(T, 1, 2))
        _, ni, nj = Z.shape
        offset = 1 if self.itself else 0
        li = torch.tensor([i for i in range(ni) for j in range(i + offset)])
        lj = torch.tensor([j for i in range(nj) for j in range(i + offset)])
        Zflat = Z[:, li, lj]
        return torch.cat([x] + [Zflat], dim=1)

    @torch.no_grad()
    def step(self, lS_o, lS_i, x):
        return self.interact_features(x, self.apply_emb(lS_o, lS_i))


def collate_criteo_offset(x_int, x_cat):
    """Reference dlrm_data_pytorch.py:397-410 (collate_wrapper_criteo_offset) on a raw batch -- x_int (B, n_dense) integer counts,
    x_cat (B, T) integer ids, as CriteoDataset.__getitem__ yields them (:372-395) -- without the click column: X = log(x_int
    as fp32 + 1), lS_o = arange(B) per table, lS_i = x_cat transposed, both (T, B) int64.  Checked against the golden vectors of
    tests/golden/collate_criteo.npz (the reference's own function); the checker of the device-side collate."""
    X_int = torch.log(torch.as_tensor(x_int).to(torch.float) + 1)
    X_cat = torch.as_tensor(x_cat).to(torch.long)
    B, T = X_cat.shape
    lS_i = torch.stack([X_cat[:, i] for i in range(T)])
    lS_o = torch.stack([torch.arange(B) for _ in range(T)])
    return X_int, lS_o, lS_i


def transform_features_terabyte(rec, max_ind_range=-1):
    """Reference script/data_loader_terabyte.py:68-87 (_transform_features) over a (B, 40) int32 block of the binary dataset as
    CriteoBinDataset.__getitem__ slices it (:226-236: column 0 the label, 1..13 the counts, 14..39 the ids), without the label:
    -> X fp32 (B, 13), lS_o (26, B), lS_i (26, B) int64.  Checked against tests/golden/collate_terabyte.npz."""
    t = torch.as_tensor(rec).view((-1, 40))
    x_int, x_cat = t[:, 1:14], t[:, 14:]
    if max_ind_range > 0:
        x_cat = x_cat % max_ind_range
    X = torch.log(x_int.clone().detach().type(torch.float) + 1)
    x_cat = x_cat.clone().detach().type(torch.long)
    B, T = x_cat.shape
    lS_o = torch.arange(B).reshape(1, -1).repeat(T, 1)
    return X, lS_o, x_cat.t().contiguous()
