"""Device-resident batch loader standing in for ``torch.utils.data.DataLoader`` in the training entry point.

The reference feeds training through a DataLoader whose dataset encodes the conditions of ONE item per call
in 8 worker processes (scripts/training.py:17, src/datasets.py:20-23).  Here conditions are encoded once for the
whole set, inputs and conditions live on the GPU, and a batch is an index gather on the device (sampled Dirichlet
conditions are drawn there too, ``datasets.DeviceConditionEncoder``).  The shuffled
index stream is generated exactly like ``RandomSampler`` does it (a seed drawn from the default torch
generator, then ``randperm`` from a generator with that seed), so a given ``torch.manual_seed`` yields the same
batches as the reference.  Under data parallelism every rank walks the same global stream and takes its
contiguous slice of each global batch.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


def shard_range(n: int, rank: int, world: int):
    """Contiguous [begin, end) share of n items for ``rank`` of ``world`` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    b = rank * base + min(rank, extra)
    return b, b + base + (1 if rank < extra else 0)


class DeviceBatchLoader:
    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, device="cuda:0", rank=0, world=1):
        self.dataset = dataset
        self.batch_sampler = SimpleNamespace(batch_size=batch_size)   # what fit() reads for its progress bar
        self.drop_last = drop_last
        self.shuffle, self.device, self.rank, self.world = shuffle, device, rank, world
        cond = dataset.conditioner
        self.random = cond.is_random() if hasattr(cond, "is_random") else False
        self.x = torch.from_numpy(np.ascontiguousarray(dataset.inputs, dtype=np.float32)).to(device)
        from .datasets import DeviceConditionEncoder
        self.enc = DeviceConditionEncoder(cond, dataset.conditions, device)

    def __len__(self):
        n, gb = len(self.dataset), self.batch_sampler.batch_size * self.world
        if self.drop_last:
            return n // gb
        return n // gb + (1 if (n % gb) // self.world > 0 else 0)

    def _order(self):
        n = len(self.dataset)
        # a DataLoader iterator first draws a base seed for its workers, then RandomSampler draws its own seed
        torch.empty((), dtype=torch.int64).random_()
        if not self.shuffle:
            return torch.arange(n)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g)

    def __iter__(self):
        order = self._order()
        B = self.batch_sampler.batch_size
        gb = B * self.world
        n = order.numel()
        for g0 in range(0, n, gb):
            glob = order[g0:g0 + gb]
            if glob.numel() < gb and self.drop_last:
                break
            if self.world > 1 and glob.numel() % self.world:
                # a ragged last global batch: every rank must get the same number of rows (the gradient average and
                # cross-rank BatchNorm statistics weigh ranks equally, and no rank may be left without rows), so the
                # last (fewer than world) indices of the epoch are dropped
                glob = glob[:glob.numel() - glob.numel() % self.world]
                if glob.numel() == 0:
                    break
            b, e = shard_range(glob.numel(), self.rank, self.world)
            idx = glob[b:e]
            idx_d = idx.to(self.device)
            x = self.x.index_select(0, idx_d)
            c = self.enc.encode(idx_d, self.dataset.num_samples)
            if c.ndim == 3 and c.shape[1] == 1:
                c = c[:, 0].contiguous()        # one draw per item collates to [B, C]: the dataset squeezes it (src/datasets.py:22)
            yield x, c
