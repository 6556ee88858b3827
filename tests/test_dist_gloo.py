"""CPU, world_size 2 over gloo: the multi-GPU plumbing of the calibration path — batch sharding,
all-gather of the cached activation slabs in batch order, broadcast of learned parameters."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "eda-dm_amd"))
    from edadm import dist as ed
    n_batches = 5
    mine = ed.shard_batches(n_batches)
    local = {i: torch.full((3, 4), float(i)) + torch.arange(4.0) for i in mine}
    full = ed.all_gather_batches(local, n_batches)
    ok = len(full) == n_batches and all(torch.equal(full[i], torch.full((3, 4), float(i)) + torch.arange(4.0))
                                        for i in range(n_batches))
    p = torch.nn.Parameter(torch.full((7,), float(rank + 1)))
    ed.broadcast_params([p])
    ok = ok and bool((p.data == 1.0).all())
    ret[rank] = (ok, mine)
    dist.destroy_process_group()


def test_shard_gather_broadcast_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == [0, 1, 2] and ret[1][1] == [3, 4]          # contiguous blocks: the gathered slab is in batch order


def test_single_process_is_identity():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "eda-dm_amd"))
    from edadm import dist as ed
    assert ed.world() == (0, 1)
    assert ed.shard_batches(3) == [0, 1, 2]
    loc = {i: torch.ones(2) * i for i in range(3)}
    assert [float(t[0]) for t in ed.all_gather_batches(loc, 3)] == [0.0, 1.0, 2.0]
