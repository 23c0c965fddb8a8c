"""Data-parallel plumbing over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in CPU tests).

The path shards by data.  Training: every rank holds the full 41.6 MB of weights and a slice of the global batch;
the flat gradient arena is all-reduced in two spans (decoder first, while the encoder backward still runs), then
averaged, before the fused clip + Adam step — so clipping sees the same averaged gradient a single-device step on
the global batch would.  Scoring / imputation: profiles are split into contiguous per-rank ranges and the only
collective is one reduce of (sum of log-likelihoods, count).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .loaders import shard_range  # noqa: F401  (re-exported)


class GradSync:
    """Asynchronous sum all-reduce of arena spans; ``finish`` waits for all of them.  The caller divides by
    ``world`` afterwards (on the GPU with the library's scale kernel)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._handles: List = []

    def launch(self, span: torch.Tensor):
        if self.world > 1:
            self._handles.append(dist.all_reduce(span, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for h in self._handles:
            h.wait()
        self._handles = []

    def sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def reduce_loglik(local_sum: float, local_count: int, device="cpu", group=None):
    """One collective for sharded scoring: global (sum of log-likelihoods, number of profiles)."""
    t = torch.tensor([float(local_sum), float(local_count)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0].item()), int(round(t[1].item()))
