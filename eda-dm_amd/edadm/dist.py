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
    """Calibration batches owned by `rank`: one contiguous block of ceil(n / world) indices per rank (the last ranks may
    own fewer, or none), so that the all-gathered slab IS the cache in batch order -- no re-ordering copy."""
    if rank is None:
        rank, size = world()
    per = (n_batches + size - 1) // size
    return list(range(min(rank * per, n_batches), min((rank + 1) * per, n_batches)))


def shard_round_robin(n, rank=None, size=None):
    """Sampling batches owned by `rank`: i with i % world == rank (rank r generates batches
    {i : i mod world = r} of the reference's batch sequence)."""
    if rank is None:
        rank, size = world()
    return [i for i in range(n) if i % size == rank]


def gather_rows(local, n_batches, group=None):
    """local: {batch index: tensor [rows, ...]} for this rank's block -> ONE tensor [n_batches * rows, ...] holding all
    batches in index order on every rank.  Each rank writes its batches straight into its slot of the output and one
    all_gather_into_tensor (RCCL: every GPU has a direct xGMI link to each peer) fills the rest in place: peak memory is
    the cache itself, not twice it.  Every rank must own at least one batch (checked before any collective, on all
    ranks alike, so a bad configuration raises everywhere instead of hanging the ranks that did enter)."""
    rank, size = world()
    if size == 1:
        return torch.cat([local[i] for i in range(n_batches)])
    per = (n_batches + size - 1) // size
    if (size - 1) * per >= n_batches:
        raise ValueError("%d calibration batches cannot be sharded over %d ranks (a rank would own none)" % (n_batches, size))
    ref = next(iter(local.values()))
    rows = ref.shape[0]
    out = torch.empty((size * per * rows,) + tuple(ref.shape[1:]), dtype=ref.dtype, device=ref.device)
    slot = out[rank * per * rows:(rank + 1) * per * rows]
    mine = shard_batches(n_batches, rank, size)
    for j, i in enumerate(mine):
        slot[j * rows:(j + 1) * rows].copy_(local[i])
    if len(mine) < per:
        slot[len(mine) * rows:].zero_()                       # the padding of the last block travels too
    if dist.get_backend(group) == "gloo":
        # CPU tests, and the 2-processes-on-one-GPU test (RCCL refuses two ranks per device): gloo gathers host tensors
        parts = [torch.empty(slot.shape, dtype=slot.dtype) for _ in range(size)]
        dist.all_gather(parts, slot.detach().cpu(), group=group)
        for r in range(size):
            if r != rank:
                out[r * per * rows:(r + 1) * per * rows].copy_(parts[r])
    else:
        dist.all_gather_into_tensor(out, slot, group=group)
    GATHER_STATS["bytes"] += out.numel() * out.element_size()
    GATHER_STATS["calls"] += 1
    return out[:n_batches * rows]


GATHER_STATS = {"bytes": 0, "calls": 0}      # bytes of gathered slabs (bench.py's multi-rank calibration leg)


def all_gather_batches(local, n_batches, group=None):
    """local: {batch index: tensor}; returns the list of all n_batches tensors in index order on every rank (views of
    the one gathered slab)."""
    if world()[1] == 1:
        return [local[i] for i in range(n_batches)]
    full = gather_rows(local, n_batches, group)
    rows = full.shape[0] // n_batches
    return [full[i * rows:(i + 1) * rows] for i in range(n_batches)]


def broadcast_params(tensors, src=0):
    _, size = world()
    if size == 1:
        return
    gloo = dist.get_backend() == "gloo"
    for t in tensors:
        d = t.data if hasattr(t, "data") else t
        if gloo and d.is_cuda:                      # see gather_rows
            h = d.detach().cpu()
            dist.broadcast(h, src=src)
            d.copy_(h)
        else:
            dist.broadcast(d, src=src)
