"""Data-parallel plumbing over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in CPU tests).

The path shards by data.  Training: every rank holds the full 41.6 MB of weights and a slice of the global batch;
the flat gradient arena is all-reduced in four spans as their gradients become final (each network's heads, then
its trunk; everything but the encoder trunk overlaps later backward work), then averaged, before the fused clip + Adam step — so clipping sees the same averaged gradient a single-device step on
the global batch would.  Scoring / imputation: profiles are split into contiguous per-rank ranges and the only
collective is one reduce of (sum of log-likelihoods, count); when there are too few profiles per GPU to fill the
machine the Monte-Carlo draws can be split instead, each rank keeping a running (max, sum of exp) per profile that two
all-reduces combine (``combine_lse``).

Ordering of the asynchronous gradient all-reduce on RCCL: ``dist.all_reduce(..., async_op=True)`` makes RCCL's own
stream wait on an event recorded on torch's current stream at the moment of the call, and the library's kernels are
launched on that same current stream, so the reduction of a span starts after every kernel that wrote it; kernels
enqueued later only write other spans (a network's heads and trunk occupy disjoint ranges of the arena, and gradients
are overwritten once per step, never touched again before ``finish``).  ``finish()`` (``Work.wait``) makes the current
stream wait for the reductions before the scale kernel and the optimiser read the arena, and the next step's backward
is enqueued behind those on the same stream.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .loaders import shard_range  # noqa: F401  (re-exported)


def _staged(group) -> bool:
    """gloo moves host memory: device tensors are staged through the host (tests on one GPU); RCCL takes them as is."""
    return dist.get_backend(group) == "gloo"


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    if t.is_cuda and _staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_max_(t: torch.Tensor, group=None) -> torch.Tensor:
    if t.is_cuda and _staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def combine_lse(m_run: torch.Tensor, s_run: torch.Tensor, group=None):
    """Merge per-rank running log-sum-exp states over disjoint sets of Monte-Carlo draws: every rank holds, per
    profile, m = max of its log-weights and s = sum of exp(log-weight - m).  One all-reduce(max) gives the global
    maximum, one all-reduce(sum) of the rescaled sums the global sum; log-sum-exp over all draws is m + log(s)
    (the combination the importance-weighted estimate of src/testing_lib.py:14 needs when K is split).  A rank with no
    draws passes m = -inf, s = 0.  In place; returns (m, s)."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return m_run, s_run
    local_m = m_run.clone()
    all_reduce_max_(m_run, group)
    s_run.mul_(torch.where(torch.isfinite(local_m), torch.exp(local_m - m_run), torch.zeros_like(s_run)))
    all_reduce_sum_(s_run, group)
    return m_run, s_run


def all_gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """[world, *local.shape] with every rank's ``local`` (contiguous)."""
    world = dist.get_world_size(group)
    if local.is_cuda and _staged(group):
        parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(world)]
        dist.all_gather(parts, local.cpu(), group=group)
        return torch.stack(parts).to(local.device)
    out = torch.empty((world, *local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(world * local.shape[0], *local.shape[1:]), local.contiguous(), group=group)
    return out


class BatchNormSync:
    """Cross-rank BatchNorm statistics (``torch.nn.SyncBatchNorm`` semantics): with it, N ranks on 1/N of the batch
    each compute what one device computes on the whole batch.  Every rank must hold the same number of rows."""

    def __init__(self, group=None, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # ``force``: run the collectives at world size 1 as well (identities there): exercises the RCCL calls and their
        # ordering against the library's kernels on a single GPU
        self.active = dist.is_initialized() and (self.world > 1 or force)

    def gather_stats(self, local: torch.Tensor) -> torch.Tensor:      # [2, N] -> [world, 2, N]
        return all_gather_rows(local, self.group)

    def sum_(self, sums: torch.Tensor) -> torch.Tensor:               # [2, N] in place
        return all_reduce_sum_(sums, self.group)


class GradSync:
    """Asynchronous sum all-reduce of arena spans; ``finish`` waits for all of them.  The caller divides by
    ``world`` afterwards (on the GPU with the library's scale kernel)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, force: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)     # ``force``: see BatchNormSync
        self._handles: List = []

    def launch(self, span: torch.Tensor):
        if self.active:
            if span.is_cuda and _staged(self.group):
                all_reduce_sum_(span, self.group)
            else:
                self._handles.append(dist.all_reduce(span, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for h in self._handles:
            h.wait()
        self._handles = []

    def sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self.active:
            all_reduce_sum_(t, self.group)
        return t


def reduce_loglik(local_sum: float, local_count: int, device="cpu", group=None):
    """One collective for sharded scoring: global (sum of log-likelihoods, number of profiles)."""
    t = torch.tensor([float(local_sum), float(local_count)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0].item()), int(round(t[1].item()))
