"""Dataset wrappers (reference: ``src/datasets.py``) plus a vectorised feed.

``ConditionedDataset`` keeps the reference's per-item interface so existing ``DataLoader`` code works.
``ConditionFeed`` is what the HIP path's own drivers use: conditions are encoded once for the whole
array (or once per batch when a Dirichlet tag samples), so the host never runs a per-item Python
transform while the GPU waits.
"""
from __future__ import annotations

import numpy as np
import torch


class ConditionedDataset(torch.utils.data.Dataset):
    def __init__(self, inputs, conditions=None, conditioner=None, num_samples=1):
        self.inputs = inputs
        self.conditions = conditions      # raw, per tag
        self.conditioner = conditioner
        self.num_samples = num_samples

    def __len__(self):
        return self.inputs.shape[0]

    def __getitem__(self, idx):
        raw = {k: v[[idx]] for k, v in self.conditions.items()}
        cond = self.conditioner.transform(raw, num_samples=self.num_samples).squeeze()
        return torch.tensor(self.inputs[idx]).float(), torch.tensor(cond).float()


class ConditionFeed:
    """Batched (inputs, conditions) iterator over contiguous index ranges.

    Deterministic conditioners are encoded once up front; sampling conditioners are encoded per batch with
    ``num_samples`` draws and returned as ``[batch, num_samples, C]`` — the layout a ``DataLoader`` over
    ``ConditionedDataset`` would have collated."""

    def __init__(self, inputs, conditions, conditioner, batch_size, num_samples=1, device="cpu"):
        self.inputs, self.conditions, self.conditioner = inputs, conditions, conditioner
        self.batch_size, self.num_samples, self.device = batch_size, num_samples, device
        self.random = conditioner.is_random() if hasattr(conditioner, "is_random") else (
            "dir" in conditioner.types and conditioner.transformers[conditioner.tags[conditioner.types.index("dir")]].transform_style == "sample")
        self._all = None
        if not self.random:
            self._all = np.asarray(conditioner.transform(conditions), dtype=np.float32)

    def __len__(self):
        return (self.inputs.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.inputs.shape[0]
        for b0 in range(0, n, self.batch_size):
            b1 = min(n, b0 + self.batch_size)
            x = torch.from_numpy(np.ascontiguousarray(self.inputs[b0:b1], dtype=np.float32))
            if self.random:
                raw = {k: v[b0:b1] for k, v in self.conditions.items()}
                c = np.asarray(self.conditioner.transform(raw, num_samples=self.num_samples), dtype=np.float32)
                c = torch.from_numpy(np.ascontiguousarray(c.transpose(1, 0, 2)))
            else:
                c = torch.from_numpy(self._all[b0:b1])
            yield b0, b1, x.to(self.device), c.to(self.device)
