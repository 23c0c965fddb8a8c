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


class DeviceConditionEncoder:
    """Conditions of a whole set held on the GPU in encoded form; a batch is a row gather.

    Deterministic tags are encoded once by the conditioner's own transformers.  A sampling Dirichlet tag
    (``DirichletTransformer`` with style "sample", src/utils.py:112-129) keeps its concentration vectors on the
    device and draws ``num_samples`` fresh Dirichlet vectors per row and call with the HIP kernel, written in
    place between the other tags' columns: the result is the ``[batch, num_samples, C]`` tensor a ``DataLoader``
    over ``ConditionedDataset`` collates (src/conditioning_lib.py:218-229, src/datasets.py:20-23).  The draws come
    from Philox counters seeded from torch's CPU generator, not from numpy's global stream."""

    def __init__(self, conditioner, conditions, device):
        from . import _native as nv
        self.nv, self.device = nv, torch.device(device)
        self.random = conditioner.is_random()
        self.cond_dim = conditioner.cond_dim
        self.gamma, self.col0, self.T = None, 0, 0
        blocks, col = [], 0
        for tag in conditioner.tags:
            tr = conditioner.transformers[tag]
            if self.random and tag == "users":
                g = np.ascontiguousarray(conditions[tag], dtype=np.float32)
                self.gamma, self.col0, self.T = torch.from_numpy(g).to(self.device), col, g.shape[1]
                blocks.append(np.zeros(g.shape, dtype=np.float32))
                col += g.shape[1]
            else:
                enc = np.asarray(tr.transform(conditions[tag]), dtype=np.float32)
                blocks.append(enc)
                col += enc.shape[1]
        self.table = torch.from_numpy(np.ascontiguousarray(np.concatenate(blocks, axis=-1))).to(self.device)
        self._calls = 0
        self._seed = None

    def encode(self, idx: torch.Tensor, num_samples: int = 1) -> torch.Tensor:
        """``idx``: int64 row indices on the device.  Returns [B, C] (deterministic) or [B, S, C]."""
        base = self.table.index_select(0, idx)
        if not self.random:
            return base
        nv = self.nv
        if self.device.type != "cuda":
            raise RuntimeError("sampled Dirichlet conditions are drawn by the HIP feed kernel: put the encoder on the GPU "
                               "(a host-side DataLoader over ConditionedDataset still samples with numpy)")
        B, S = idx.numel(), int(num_samples)
        out = base.unsqueeze(1).expand(B, S, self.cond_dim).contiguous()
        if self._seed is None:
            self._seed = int(torch.empty((), dtype=torch.int64).random_().item())
        nv.ensure_device(self.device)
        nv.call("btvae_dirichlet_sample", nv.ptr(self.gamma), nv.ptr(idx.contiguous()), nv.ptr(out), B, S, self.T, self.cond_dim,
                self.col0, self._seed, 64 * self._calls)
        self._calls += 1
        return out


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
        self._dev_enc = None
        if not self.random:
            self._all = np.asarray(conditioner.transform(conditions), dtype=np.float32)
        elif torch.device(device).type == "cuda":
            self._dev_enc = DeviceConditionEncoder(conditioner, conditions, device)

    def __len__(self):
        return (self.inputs.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.inputs.shape[0]
        for b0 in range(0, n, self.batch_size):
            b1 = min(n, b0 + self.batch_size)
            x = torch.from_numpy(np.ascontiguousarray(self.inputs[b0:b1], dtype=np.float32))
            if self._dev_enc is not None:
                c = self._dev_enc.encode(torch.arange(b0, b1, device=self.device), self.num_samples)
                yield b0, b1, x.to(self.device), c
                continue
            if self.random:
                raw = {k: v[b0:b1] for k, v in self.conditions.items()}
                c = np.asarray(self.conditioner.transform(raw, num_samples=self.num_samples), dtype=np.float32)
                c = torch.from_numpy(np.ascontiguousarray(c.transpose(1, 0, 2)))
            else:
                c = torch.from_numpy(self._all[b0:b1])
            yield b0, b1, x.to(self.device), c.to(self.device)
