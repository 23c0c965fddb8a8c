"""Condition vectors fed to the conditional VAE (reference: ``src/conditioning_lib.py``).

This is the *feed* of the hot path, not a kernel: it stays numpy on the host and produces the dense
``c[N,C]`` (or ``c[S,N,C]`` for sampled Dirichlet user features) layout the model consumes.  The
per-tag encoders are: circ -> (cos, sin), cat -> one-hot, cont -> quantile-normal, ord -> ordinal,
dir -> Dirichlet features of per-user concentration vectors, identity -> as is.  Fitting the user
topic model itself (sklearn LDA, ``src/user_encoding_lib.py``) is outside this package: per-user
concentration matrices are taken as inputs.
"""
from __future__ import annotations

import os
import pickle

import numpy as np

from .utils import CircularTransformer, DirichletTransformer, IdentityTransformer


def _calendar(raw_dates, fn):
    if raw_dates is None:
        raise ValueError("Raw dates must be provided.")
    return np.array([fn(d) for d in raw_dates])


def _register(kw, cset, tag, typ, support, values):
    kw["tags"].append(tag); kw["types"].append(typ); kw["supports"].append(support)
    cset[tag] = values


def add_months(condition_kwargs, condition_set, raw_dates=None):
    v = _calendar(raw_dates, lambda d: d.month)
    _register(condition_kwargs, condition_set, "months", "circ", np.unique(v).tolist(), v[..., None])


def add_weekdays(condition_kwargs, condition_set, raw_dates=None):
    v = _calendar(raw_dates, lambda d: d.weekday())
    _register(condition_kwargs, condition_set, "weekdays", "circ", np.unique(v).tolist(), v[..., None])


def add_years(condition_kwargs, condition_set, raw_dates=None):
    v = _calendar(raw_dates, lambda d: d.year)
    _register(condition_kwargs, condition_set, "years", "ord", np.unique(v).tolist(), v[..., None])


def add_is_weekend(condition_kwargs, condition_set, raw_dates=None):
    v = _calendar(raw_dates, lambda d: d.weekday() >= 5)
    _register(condition_kwargs, condition_set, "is_weekend", "cat", [0, 1], v[..., None])


def _shifted(data, days, fill_nan=True):
    """Profiles of ``days`` days earlier for every user-day, first ``days`` days NaN (circular shift)."""
    head = data[:, -days:, :] * (np.nan if fill_nan else 1.0)
    return np.concatenate([head, data[:, :-days, :]], axis=1).reshape(-1, data.shape[-1])


def add_day_befores(condition_kwargs, condition_set, data=None):
    if data is None: raise ValueError("Data must be provided.")
    v = _shifted(data, 1)
    _register(condition_kwargs, condition_set, "day_befores", "identity", [np.nanmin(v), np.nanmax(v)], v)


def add_twoday_befores(condition_kwargs, condition_set, data=None):
    if data is None: raise ValueError("Data must be provided.")
    v = _shifted(data, 2)
    _register(condition_kwargs, condition_set, "twoday_befores", "identity", [np.nanmin(v), np.nanmax(v)], v)


def add_week_befores(condition_kwargs, condition_set, data=None):
    if data is None: raise ValueError("Data must be provided.")
    v = _shifted(data, 7)
    _register(condition_kwargs, condition_set, "week_befores", "identity", [np.nanmin(v), np.nanmax(v)], v)


def add_month_befores(condition_kwargs, condition_set, data=None):
    if data is None: raise ValueError("Data must be provided.")
    v = _shifted(data, 28, fill_nan=False)
    _register(condition_kwargs, condition_set, "month_befores", "identity", [np.nanmin(v), np.nanmax(v)], v)


def add_absmean(condition_kwargs, condition_set, data=None):
    if data is None: raise ValueError("Data must be provided.")
    v = np.nanmean(np.abs(data), axis=(1, 2), keepdims=True)[..., 0]
    _register(condition_kwargs, condition_set, "absmean", "cont", [np.nanmin(v), np.nanmax(v)], v.repeat(data.shape[1], axis=0))


def add_users(condition_kwargs, condition_set, data=None, user_gamma=None, doc_topic_prior=0.01, max_doc_length=None):
    """Per-user Dirichlet concentrations repeated for every day (reference: src/conditioning_lib.py:71-99).
    ``user_gamma[num_users, T]`` must be supplied (e.g. a cached ``user_gamma.npy``)."""
    if data is None: raise ValueError("Data must be provided.")
    if user_gamma is None:
        raise NotImplementedError("fitting the user topic model is outside this package: pass user_gamma")
    hi = max_doc_length if max_doc_length is not None else float(user_gamma.sum(1).max())
    _register(condition_kwargs, condition_set, "users", "dir", [doc_topic_prior, hi], user_gamma.repeat(data.shape[1], axis=0))


def prepare_conditions(condition_tag_list, raw_dates=None, data=None, missing_data=None, user_gamma=None, **_):
    kw = {"tags": [], "types": [], "supports": []}
    cset = {}
    date_tags = {"months": add_months, "weekdays": add_weekdays, "years": add_years, "is_weekend": add_is_weekend}
    data_tags = {"day_befores": add_day_befores, "twoday_befores": add_twoday_befores, "week_befores": add_week_befores,
                 "month_befores": add_month_befores}
    for tag in condition_tag_list:
        if tag in date_tags:
            date_tags[tag](kw, cset, raw_dates)
        elif tag in data_tags:
            data_tags[tag](kw, cset, data)
        elif tag == "absmean":
            add_absmean(kw, cset, missing_data)
        elif tag == "users":
            add_users(kw, cset, missing_data, user_gamma=user_gamma)
        else:
            raise ValueError("Unknown condition tag.")
    return kw, cset


class Conditioner:
    """One transformer per tag; ``transform`` concatenates the encoded features in tag order
    (reference: src/conditioning_lib.py:176-255)."""

    def __init__(self, tags, supports, types, condition_set=None):
        self.tags, self.supports, self.types = tags, supports, types
        self.cond_dim = 0
        self.transformers = {}
        self.init_transformers(condition_set)

    def add_transformer(self, tag, support, typ, data=None):
        if typ == "circ":
            self.transformers[tag] = CircularTransformer(max_conds=np.max(support), min_conds=np.min(support))
            self.cond_dim += 2
        elif typ == "cat":
            from sklearn.preprocessing import OneHotEncoder
            self.transformers[tag] = OneHotEncoder(sparse_output=False).fit(data)
            self.cond_dim += self.transformers[tag].categories_[0].shape[0]
        elif typ == "cont":
            from sklearn.preprocessing import QuantileTransformer
            self.transformers[tag] = QuantileTransformer(output_distribution="normal").fit(data)
            self.cond_dim += 1
        elif typ == "ord":
            from sklearn.preprocessing import OrdinalEncoder
            self.transformers[tag] = OrdinalEncoder(categories=[support]).fit(data)
            self.cond_dim += 1
        elif typ == "dir":
            self.transformers[tag] = DirichletTransformer(gammas=data)
            self.cond_dim += data.shape[-1]
        elif typ == "identity":
            self.transformers[tag] = IdentityTransformer()
            self.cond_dim += data.shape[-1]
        else:
            raise ValueError("Unknown type.")

    def init_transformers(self, data):
        for tag, support, typ in zip(self.tags, self.supports, self.types):
            self.add_transformer(tag, support, typ, data[tag])

    def add_condition(self, tag, support, typ, data=None):
        self.tags.append(tag); self.supports.append(support); self.types.append(typ)
        self.add_transformer(tag, support, typ, data)

    def is_random(self) -> bool:
        """True when a Dirichlet tag draws fresh samples on every ``transform`` call."""
        return "dir" in self.types and self.transformers[self.tags[self.types.index("dir")]].transform_style == "sample"

    def transform(self, data, num_samples=1):
        parts = []
        if self.is_random():
            for tag in self.tags:
                if tag == "users":
                    parts.append(self.transformers[tag].transform(data[tag], num_samples=num_samples))
                else:
                    parts.append(np.repeat(self.transformers[tag].transform(data[tag])[None, ...], num_samples, axis=0))
        else:
            parts = [self.transformers[tag].transform(data[tag]) for tag in self.tags]
        return np.concatenate(parts, axis=-1)

    def save(self, save_path):
        with open(os.path.join(save_path, 'conditioner.pkl'), 'wb') as f:
            pickle.dump(self, f)
        print(f"Conditioner saved to {save_path}")

    @staticmethod
    def load(folder_path):
        with open(os.path.join(folder_path, 'conditioner.pkl'), 'rb') as f:
            return pickle.load(f)
