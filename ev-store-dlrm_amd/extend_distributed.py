"""Mirror of the reference's distributed shim for the inference forward path.

Reference: extend_distributed.py -- get_my_slice (:47-51), get_split_lengths (:54-62),
alltoall / All2All_Req.forward / All2All_Wait.forward (:541-576, :389-426, :444-465),
all_gather (:579-...), barrier (:587).  Forward only (inference); one process per GPU,
torch.distributed "nccl" backend (= RCCL over xGMI on MI355X) or "gloo" on CPU.
"""
import os

import torch
import torch.distributed as dist

my_size = 1
my_rank = 0
my_local_rank = 0


def init_distributed(rank=-1, local_rank=-1, size=-1, use_gpu=False, backend=""):
    """extend_distributed.py:65-191, reduced to what torchrun-style launches need."""
    global my_size, my_rank, my_local_rank
    if dist.is_available() and dist.is_initialized():
        my_size, my_rank = dist.get_world_size(), dist.get_rank()
        my_local_rank = int(os.environ.get("LOCAL_RANK", my_rank))
        return
    size = int(os.environ.get("WORLD_SIZE", size))
    if size <= 1:
        my_size, my_rank, my_local_rank = 1, 0, 0
        return
    rank = int(os.environ.get("RANK", rank))
    my_local_rank = int(os.environ.get("LOCAL_RANK", local_rank if local_rank >= 0 else rank))
    if not backend:
        backend = "nccl" if use_gpu else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if use_gpu:
        torch.cuda.set_device(my_local_rank)
    dist.init_process_group(backend, rank=rank, world_size=size)
    my_size, my_rank = size, rank


def get_my_slice(n, rank=None, size=None):
    """Contiguous slice of n items owned by this rank (extend_distributed.py:47-51)."""
    rank = my_rank if rank is None else rank
    size = my_size if size is None else size
    k, m = divmod(n, size)
    return slice(rank * k + min(rank, m), (rank + 1) * k + min(rank + 1, m), 1)


def get_split_lengths(n, rank=None, size=None):
    """(my_len, per-rank lengths or None when even) (extend_distributed.py:54-62)."""
    rank = my_rank if rank is None else rank
    size = my_size if size is None else size
    k, m = divmod(n, size)
    if m == 0:
        return k, None
    splits = [(k + 1) if i < m else k for i in range(size)]
    return splits[rank], splits


class _Request:
    """What ext_dist.alltoall returns: .wait() -> tuple of per-source-rank (B_local, T_p*d) blocks."""

    def __init__(self, work, output, table_split_lengths, local_batch_num):
        self.req, self.output = work, output
        self.table_split_lengths, self.local_batch_num = table_split_lengths, local_batch_num

    def wait(self):
        if self.req is not None:
            self.req.wait()
            self.req = None
        outs = self.output.split(self.table_split_lengths)
        return tuple(o.view([self.local_batch_num, -1]) for o in outs)


def alltoall(inputs, per_rank_table_splits, group=None):
    """Forward of ext_dist.alltoall (extend_distributed.py:541-576 -> All2All_Req :389-426).

    inputs: list of T_local (B,d) pooled tensors (full batch, local tables).
    per_rank_table_splits: tables per rank (None when even).
    Sends rows [batch slice of peer p] of cat(inputs, dim=1); receives from source rank p a
    (B_local, T_p*d) block -- "local tables x full batch" -> "all tables x local batch".
    """
    if hasattr(inputs, "materialize"):   # dlrm_ops.LazyPooled: torch.cat takes lists / tuples only
        inputs = inputs.materialize()
    batch_size, emb_dim = inputs[0].size()
    local_table_num = len(inputs)
    local_batch_num, batch_splits = get_split_lengths(batch_size)
    global_table_num = sum(per_rank_table_splits) if per_rank_table_splits else local_table_num * my_size
    in_splits = [m * emb_dim * local_table_num for m in batch_splits] if batch_splits else None
    if per_rank_table_splits:
        out_splits = [local_batch_num * e * emb_dim for e in per_rank_table_splits]
    else:
        out_splits = None
    inp = torch.cat(inputs, dim=1).view([-1])
    output = inp.new_empty([global_table_num * local_batch_num * emb_dim])
    work = dist.all_to_all_single(output, inp, out_splits, in_splits, group=group, async_op=True)
    tsl = out_splits if out_splits else local_table_num * local_batch_num * emb_dim
    return _Request(work, output, tsl, local_batch_num)


def all_gather(input, lengths, dim=0):
    """extend_distributed.py:579-584 (forward only)."""
    if my_size == 1:
        return input
    if not lengths:
        outs = [torch.empty_like(input) for _ in range(my_size)]
    else:
        shape = list(input.shape)
        outs = []
        for n in lengths:
            shape[dim] = n
            outs.append(input.new_empty(shape))
    dist.all_gather(outs, input.contiguous())
    return torch.cat(outs, dim=dim)


def barrier():
    if my_size > 1:
        dist.barrier()
