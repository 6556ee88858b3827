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


def _sample_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "eda-dm_amd"))
    from edadm.sample_driver import ShardedSampler, batch_noise

    class Loop:                                     # stands in for DDIMLoop: returns what it was given
        def sample(self, x_T, cond, uncond):
            return x_T + cond

    s = ShardedSampler(Loop(), seed=7, total_images=22, batch=4, shape=(3, 2, 2), n_classes=10, device="cpu")
    got = {}
    n = s.run(lambda i, labels: (labels.float().view(-1, 1, 1, 1), None), lambda i, img: got.__setitem__(i, img.clone()))
    ret[rank] = (s.my_batches(), n, {i: v.numpy() for i, v in got.items()},
                 batch_noise(7, 3, (4, 3, 2, 2), "cpu").numpy())
    dist.destroy_process_group()


def test_sampling_shards_are_a_function_of_seed_and_batch_index():
    """rank r generates batches {i : i mod world = r}; a batch depends on (seed, batch index) only, so two
    ranks together make exactly the images one rank makes alone."""
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "eda-dm_amd"))
    from edadm.sample_driver import ShardedSampler
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sample_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0][0] == [0, 2, 4] and ret[1][0] == [1, 3, 5] and ret[0][1] == ret[1][1] == 3
    assert np.array_equal(ret[0][3], ret[1][3])

    class Loop:
        def sample(self, x_T, cond, uncond):
            return x_T + cond

    alone = {}
    s = ShardedSampler(Loop(), seed=7, total_images=22, batch=4, shape=(3, 2, 2), n_classes=10, device="cpu")
    assert s.my_batches() == [0, 1, 2, 3, 4, 5]
    s.run(lambda i, labels: (labels.float().view(-1, 1, 1, 1), None), lambda i, img: alone.__setitem__(i, img.numpy()))
    both = dict(ret[0][2])
    both.update(ret[1][2])
    assert sorted(both) == sorted(alone) == list(range(6))
    for i in range(6):
        assert np.array_equal(both[i], alone[i]), i
    assert not np.array_equal(alone[0], alone[1])


def test_gather_refuses_a_rank_without_batches_before_any_collective():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "eda-dm_amd"))
    from edadm import dist as ed
    assert ed.shard_batches(5, 0, 4) == [0, 1] and ed.shard_batches(5, 2, 4) == [4] and ed.shard_batches(5, 3, 4) == []
    assert ed.shard_batches(8, 3, 4) == [6, 7]
    assert ed.shard_round_robin(5, 1, 2) == [1, 3]
