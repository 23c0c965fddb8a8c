"""Result files of the reference's evaluation scripts, written from streamed per-profile metrics.

``scripts/collect_testing_results.py:117-137`` of the reference writes, per run folder, ``test_results_raw.csv`` (one row
per profile: ``l_1, l_2, loglikelihood, user_id, month, set_type``) and ``test_results_aggregate.pkl``
(``{set_type: {"loglikelihood", "error_l_1", "error_l_2"}}`` of plain floats); its notebooks read those two files.
The functions here produce the same frames / files from the arrays ``testing_lib.mass_test_metrics`` returns.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict

import numpy as np
import pandas as pd

ERROR_KEYS = ("l_1", "l_2")


def test_results_frame(metrics: Dict[str, np.ndarray], user_ids, months, set_type: str = "test") -> pd.DataFrame:
    """Column order as the reference builds it: the error columns, then loglikelihood, user_id, month, set_type."""
    df = pd.DataFrame({k: np.asarray(metrics[k]) for k in ERROR_KEYS})
    df["loglikelihood"] = np.asarray(metrics["loglikelihood"])
    df["user_id"] = np.asarray(user_ids)
    df["month"] = np.asarray(months)
    df["set_type"] = set_type
    return df


def test_results_aggregate(metrics: Dict[str, np.ndarray]) -> Dict[str, float]:
    agg = {"loglikelihood": float(np.asarray(metrics["loglikelihood"]).mean())}
    for k in ERROR_KEYS:
        agg["error_" + k] = float(np.asarray(metrics[k]).mean())
    return agg


def save_test_results(folder: str, frames: Dict[str, pd.DataFrame], aggregates: Dict[str, Dict[str, float]]):
    """``frames`` / ``aggregates`` keyed by set type ("test", ...); same file names and layouts as the reference."""
    pd.concat([frames[s] for s in frames]).to_csv(os.path.join(folder, "test_results_raw.csv"), index=False)
    with open(os.path.join(folder, "test_results_aggregate.pkl"), "wb") as f:
        pickle.dump({s: dict(a) for s, a in aggregates.items()}, f)


def imputation_errors(samples: np.ndarray, target: np.ndarray) -> Dict[str, np.ndarray]:
    """Per-profile l_1 / l_2 errors of imputed windows as ``scripts/collect_imputation_results.py:104-112`` of the
    reference computes them: ``mean_k(sum_d |s[k,n,d] - t[n,d]|^p) ** (1/p)``.  samples [K,N,W], target [N,W]."""
    out = {}
    for p_val, key in ((1, "l_1"), (2, "l_2")):
        out[key] = np.mean((np.abs(samples - target[None, ...]) ** p_val).sum(axis=-1), axis=0) ** (1 / p_val)
    return out


def save_imputation_results(folder: str, window: int, frames: Dict[str, pd.DataFrame], aggregates: Dict[str, Dict[str, float]],
                            last_samples: Dict[str, np.ndarray]):
    """Files of the reference's imputation collection (``scripts/collect_imputation_results.py:139-148``) under
    ``<folder>/forecasting_results_<window>/``: ``test_results_raw.csv``, ``test_results_aggregate.pkl`` and
    ``test_results_samples.pkl`` (one imputed window per profile and set type)."""
    sub = os.path.join(folder, f"forecasting_results_{window}")
    os.makedirs(sub, exist_ok=True)
    pd.concat([frames[s] for s in frames]).to_csv(os.path.join(sub, "test_results_raw.csv"), index=False)
    with open(os.path.join(sub, "test_results_aggregate.pkl"), "wb") as f:
        pickle.dump({s: dict(a) for s, a in aggregates.items()}, f)
    with open(os.path.join(sub, "test_results_samples.pkl"), "wb") as f:
        pickle.dump({s: np.asarray(v) for s, v in last_samples.items()}, f)
    return sub


test_results_frame.__test__ = False
test_results_aggregate.__test__ = False
