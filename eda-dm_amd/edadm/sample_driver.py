"""Sharded sampling driver (the batch loop of scripts/sample_diffusion_ldm_imagenet.py:215-249 and its
launcher scripts/for_imagenet.sh:10-14 of the reference).

The reference draws every batch's start noise from one sequential generator, so which images a process makes depends on
how many batches ran before.  Here a batch is a pure function of (seed, global batch index): its noise and its class
labels come from a generator seeded with that pair, rank r of a world of N takes batches {i : i mod N = r}, and the union
over ranks is the same set of images for every N -- no collective on the data path (only an optional reduction of
counters at the end)."""
import math

import torch

from . import dist as edist

_MIX = 1_000_003


def batch_generator(seed, batch_index, device):
    g = torch.Generator(device=device)
    g.manual_seed((int(seed) * _MIX + int(batch_index)) % (2 ** 63 - 1))
    return g


def batch_noise(seed, batch_index, shape, device):
    """Start latents x_T of global batch `batch_index`."""
    return torch.randn(tuple(shape), generator=batch_generator(seed, batch_index, device), device=device)


def batch_labels(seed, batch_index, n, n_classes, device):
    """Class labels of global batch `batch_index` (sample_diffusion_ldm_imagenet.py:218-224 draws them at random)."""
    g = batch_generator(seed, batch_index, device)
    torch.randn(1, generator=g, device=device)          # decorrelate from the noise stream of the same pair
    return torch.randint(0, n_classes, (n,), generator=g, device=device)


class ShardedSampler:
    """Generates `total_images` in batches of `batch` over the ranks of the current process group.

    loop      : edadm.sampling.DDIMLoop / PLMSLoop on this rank's frozen engine
    cond_fn   : (global batch index, labels) -> (cond [B, L, D], uncond [B, L, D])   (class / text embedding)
    sink      : (global batch index, latents x_0 [B, C, H, W]) -> None                (decode / write / count)"""

    def __init__(self, loop, seed, total_images, batch, shape, n_classes=1000, device="cuda"):
        self.loop, self.seed, self.batch, self.shape = loop, seed, batch, tuple(shape)
        self.n_batches = math.ceil(total_images / batch)
        self.n_classes, self.device = n_classes, device

    def my_batches(self, rank=None, world=None):
        return edist.shard_round_robin(self.n_batches, rank, world)

    def run(self, cond_fn, sink, limit=None):
        """loop may be an edadm.sampling.InFlightSampler: its batches are sampled on alternating streams, n in flight, and a batch
        is handed to `sink` (on the current stream, after waiting for the batch's stream) once n - 1 later ones are enqueued."""
        done = 0
        flight = hasattr(self.loop, "submit")
        pend = []

        def deliver():
            j, lat, st = pend.pop(0)
            torch.cuda.current_stream().wait_stream(st)
            lat.record_stream(torch.cuda.current_stream())
            sink(j, lat)

        for i in self.my_batches():
            if limit is not None and done >= limit:
                break
            x_T = batch_noise(self.seed, i, (self.batch,) + self.shape, self.device)
            labels = batch_labels(self.seed, i, self.batch, self.n_classes, self.device)
            cond, uncond = cond_fn(i, labels)
            if flight:
                lat, st = self.loop.submit(x_T, cond, uncond)
                pend.append((i, lat, st))
                if len(pend) >= len(self.loop.loops):
                    deliver()
            else:
                sink(i, self.loop.sample(x_T, cond, uncond))
            done += 1
        while pend:
            deliver()
        # kernels are asynchronous: a failure inside one (a persistent-GEMM hand-off that timed out) surfaces here, where the
        # run synchronises anyway, as an exception instead of silently wrong images
        if torch.device(self.device).type == "cuda":
            from . import ops
            ops.device_status()
        return done
