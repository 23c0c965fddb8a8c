"""Synthetic daily load profiles of the shape the shipped configs produce.

The reference's data preparation (CSV loading, amputation, log normalisation) needs a dataset that is not
part of its repository; this module stands in for it with seeded synthetic data so the training entry point,
tests and benchmarks can run: profiles ~ clamp(N(1,1), -3, 5) with 10 % of entries at the zero id -3 (the range
of the log-normalised data), calendar conditions months ~ U{1..12}, weekdays ~ U{0..6}.
"""
from __future__ import annotations

import numpy as np

from .conditioning_lib import Conditioner
from .datasets import ConditionedDataset


def profiles(n, dim, rng):
    x = np.clip(rng.normal(1.0, 1.0, size=(n, dim)), -3, 5)
    x[rng.random((n, dim)) < 0.1] = -3.0
    return x.astype(np.float32)


def calendar_conditions(n, rng):
    return {"months": rng.integers(1, 13, size=(n, 1)), "weekdays": rng.integers(0, 7, size=(n, 1))}


def calendar_conditioner(condition_set):
    return Conditioner(tags=["months", "weekdays"], supports=[list(range(1, 13)), list(range(7))], types=["circ", "circ"],
                       condition_set=condition_set)


def prepare_synthetic_data(data_cfg, n_train=65536, n_val=8192):
    """(trainset, valset, conditioner) for a config's ``data`` section: input width 24 h / resolution plus padding."""
    seed = int(data_cfg.get("random_seed", 2112))
    rng = np.random.default_rng(seed)
    pad = data_cfg.get("pad", 0)
    width = 24 // int(data_cfg.get("resolution", 1)) + (sum(pad) if isinstance(pad, (list, tuple)) else int(pad))
    tags = data_cfg.get("condition_tag_list", ["months", "weekdays"])
    if sorted(tags) != ["months", "weekdays"]:
        raise NotImplementedError("synthetic data provides the calendar conditions months + weekdays only")
    ctr, cva = calendar_conditions(n_train, rng), calendar_conditions(n_val, rng)
    cond = calendar_conditioner(ctr)
    return (ConditionedDataset(profiles(n_train, width, rng), ctr, cond),
            ConditionedDataset(profiles(n_val, width, rng), cva, cond), cond)
