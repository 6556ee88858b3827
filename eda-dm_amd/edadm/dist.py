"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Sampling shards independent image batches with no collective.  Calibration shards the rows of
`save_inp_oup_data` (the O(units^2) prefix forwards) across ranks and all-gathers the cached
(inp_q, inp_fp, out_fp) slabs — the "shared FP32 reference activations" — then broadcasts the
learned parameters from rank 0 after each unit so replicas stay bit-identical."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_batches(n_batches, rank=None, size=None):
    """Batch indices owned by `rank`: i with i % world == rank (stable, order preserving)."""
    if rank is None:
        rank, size = world()
    return [i for i in range(n_batches) if i % size == rank]


def all_gather_batches(local, n_batches, group=None):
    """local: {batch index: tensor}; returns the list of all n_batches tensors in index order on
    every rank.  All tensors share a shape; ranks may own different numbers of batches."""
    rank, size = world()
    if size == 1:
        return [local[i] for i in range(n_batches)]
    per_rank = (n_batches + size - 1) // size
    ref = next(iter(local.values())) if local else None
    shape_t = torch.tensor(list(ref.shape) if ref is not None else [0], device=ref.device if ref is not None else "cpu")
    if ref is None:
        raise RuntimeError("a rank without any calibration batch cannot infer the slab shape; use n_batches >= world")
    mine = shard_batches(n_batches, rank, size)
    slab = torch.zeros((per_rank,) + tuple(ref.shape), dtype=ref.dtype, device=ref.device)
    for j, i in enumerate(mine):
        slab[j] = local[i]
    out = [torch.empty_like(slab) for _ in range(size)]
    dist.all_gather(out, slab, group=group)
    res = [None] * n_batches
    for r in range(size):
        for j, i in enumerate(shard_batches(n_batches, r, size)):
            res[i] = out[r][j]
    return res


def broadcast_params(tensors, src=0):
    _, size = world()
    if size == 1:
        return
    for t in tensors:
        dist.broadcast(t.data if hasattr(t, "data") else t, src=src)
