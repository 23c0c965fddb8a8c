"""Math helpers, condition transformers and training utilities with the reference's names
(``src/utils.py``).  The element-wise helpers below run as plain device tensor arithmetic; they serve
the API surface (arbitrary priors, notebooks) — the training and scoring paths use the fused HIP
kernels instead.
"""
from __future__ import annotations

import argparse
import math

import numpy as np
import torch


# ----- positive reparameterisations (src/utils.py:10-16) -----
def to_positive(xtilde):
    return torch.nn.functional.softplus(xtilde, beta=1, threshold=5)


to_sigma = to_positive


def matrix_normalizer(matrix):
    """Unit-norm columns (src/utils.py:18-19)."""
    return matrix / torch.linalg.norm(matrix, dim=0, ord=2, keepdim=True)


def log_prob(dist="Normal", params=None, targets=None):
    """Element-wise log density (src/utils.py:21-27)."""
    if dist != "Normal":
        raise ValueError("Unknown distribution.")
    mu, sigma = params["mu"], params["sigma"]
    return -0.5 * math.log(2 * math.pi) - torch.log(sigma) - 0.5 * ((targets - mu) / sigma) ** 2


def kl_divergence(dist="Normal", params=None, prior_params=None):
    """Element-wise KL between diagonal Gaussians (src/utils.py:29-35)."""
    if dist != "Normal":
        raise ValueError("Unknown distribution.")
    ratio = (params["sigma"] / prior_params["sigma"]).pow(2)
    shift = ((params["mu"] - prior_params["mu"]) / prior_params["sigma"]).pow(2)
    return 0.5 * (ratio + shift - 1 - ratio.log())


# ----- condition transformers (src/utils.py:82-154) -----
class IdentityTransformer:
    def fit_transform(self, data): return data
    def transform(self, data): return data
    def inverse_transform(self, data): return data


class CircularTransformer:
    """value -> (cos, sin) of its angle on a circle with ``max - min + 1`` positions."""

    def __init__(self, max_conds=None, min_conds=None):
        self.max_conds, self.min_conds = max_conds, min_conds

    def _angles(self, data):
        return (data - self.min_conds) / (self.max_conds - self.min_conds + 1) * 2 * np.pi

    def fit_transform(self, data):
        if self.max_conds is None:
            self.max_conds = np.max(data, axis=0, keepdims=False)
        if self.min_conds is None:
            self.min_conds = np.min(data, axis=0, keepdims=False)
        return self.transform(data)

    def transform(self, data):
        a = self._angles(data)
        return np.concatenate((np.cos(a), np.sin(a)), axis=1)

    def inverse_transform(self, data):
        span = self.max_conds - self.min_conds + 1
        return np.arctan2(data[:, 1], data[:, 0]) / (2 * np.pi) * span + self.min_conds


class DirichletTransformer:
    """Per-user Dirichlet concentration vectors -> sampled / mean / embedded / scaled features."""

    def __init__(self, gammas=None, transform_style="sample"):
        self.transform_style = transform_style
        self.gamma_min, self.gamma_max = gammas.min(axis=0), gammas.max(axis=0)
        totals = gammas.sum(axis=1)
        self.total_gamma_max, self.total_gamma_min = totals.max(), totals.min()

    def transform(self, gammas, transform_style=None, num_samples=1):
        if transform_style is not None:
            self.transform_style = transform_style
        style = self.transform_style
        if style == "sample":
            draws = np.random.gamma(gammas, 1, size=(num_samples, gammas.shape[0], gammas.shape[1]))
            return draws / draws.sum(-1, keepdims=True)
        total = gammas.sum(axis=1, keepdims=True)
        if style == "embed":
            out = np.zeros((gammas.shape[0], gammas.shape[1] + 1))
            out[:, :-1] = gammas / total
            out[:, -1] = (total[:, 0] - self.total_gamma_min) / (self.total_gamma_max - self.total_gamma_min + 1e-6)
            return out
        if style == "mean":
            return gammas / total
        if style == "scaled":
            return (gammas - self.gamma_min) / (self.gamma_max - self.gamma_min + 1e-6)
        raise NotImplementedError


# ----- training utilities (src/utils.py:156-215) -----
class EarlyStopping:
    """Stops after ``patience`` validations that fail to beat the best score by more than ``delta``
    (scores are maximised)."""

    def __init__(self, patience=5, delta=0.1):
        self.patience, self.delta = patience, delta
        self.best_score, self.early_stop, self.counter = None, False, 0

    def __call__(self, score, pbar=None):
        say = pbar.write if pbar is not None else print
        if self.best_score is None:
            self.best_score = score
            return
        if self.early_stop:
            return
        if score > self.best_score + self.delta:
            self.best_score, self.counter = score, 0
            say(f"New (significant) best score: {self.best_score:5e}")
            return
        self.counter += 1
        say(f'EarlyStopping counter: {self.counter} out of {self.patience}')
        if self.counter >= self.patience:
            self.early_stop = True
            say("Early stopping initiated.")


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ('yes', 'true', 't', 'y', '1'):
        return True
    if s in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')
