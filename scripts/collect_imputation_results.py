"""Forecast-by-imputation results for every trained run folder of a directory (CLI of the reference's
``scripts/collect_imputation_results.py``: ``--config_dir --overwrite --device --num_imputation_samples --batch_size
--unobserved_window_length --num_pseudo_gibbs_steps --num_metropolis_within_hasting_steps``).

Per run: the last ``unobserved_window_length`` coordinates of every test profile are hidden, imputed by pseudo-Gibbs
(+ optional Metropolis-within-Gibbs) sweeps on the device, and scored with the equal-weight mixture log-likelihood of
the held-out window; files as the reference writes them under ``<run>/forecasting_results_<window>/``.

As for ``collect_testing_results.py`` the test split comes from ``--test_npz`` (``X_test [N,D]``, ``user_ids [N]``,
``months [N]``, one ``cond_<tag> [N,.]`` per condition tag): the reference's CSV preprocessing is not part of this
package.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from behavio_temporal_vae_amd import imputation_lib, results_lib, utils  # noqa: E402
from behavio_temporal_vae_amd.conditioning_lib import Conditioner  # noqa: E402
from behavio_temporal_vae_amd.vae_models import CVAE  # noqa: E402


def run_folder(folder, data, args):
    with open(os.path.join(folder, "config.json")) as f:
        config = json.load(f)
    conditioner = Conditioner.load(folder)
    X = np.asarray(data["X_test"], dtype=np.float32)
    W = args.unobserved_window_length
    if X.shape[-1] < W:
        print(f"Unobserved window length is too long for {folder}. Skipping...")
        return None
    condition_set = {tag: data["cond_" + tag] for tag in conditioner.tags}
    model = CVAE(input_dim=X.shape[1], conditioner=conditioner, **config["model"])
    model.load(folder)
    model.eval()
    x_missing = X.copy()
    x_missing[:, -W:] = np.nan
    target = X[:, -W:]
    lls, samples = imputation_lib.mass_cvae_imputation_with_loglikelihood(
        model, x_missing, target, conditioner, condition_set, missing_data_init=None, batch_size=args.batch_size,
        num_samples=args.num_imputation_samples, warmup_steps=args.num_pseudo_gibbs_steps,
        num_iter=args.num_metropolis_within_hasting_steps, verbose_freq=100, device=args.device)
    metrics = results_lib.imputation_errors(samples, target)
    metrics["loglikelihood"] = lls
    frame = results_lib.test_results_frame(metrics, data["user_ids"], data["months"], "test")
    agg = results_lib.test_results_aggregate(metrics)
    # the ring-buffer slot the reference keeps (scripts/collect_imputation_results.py:148)
    slot = args.num_imputation_samples % (args.num_pseudo_gibbs_steps + args.num_metropolis_within_hasting_steps) - 1
    results_lib.save_imputation_results(folder, W, {"test": frame}, {"test": agg}, {"test": samples[slot]})
    return agg


def main(args):
    data = dict(np.load(args.test_npz, allow_pickle=False))
    done = {}
    sub = f"forecasting_results_{args.unobserved_window_length}"
    for i, name in enumerate(sorted(os.listdir(args.config_dir))):
        folder = os.path.join(args.config_dir, name)
        if not os.path.exists(os.path.join(folder, "trained_model.pt")):
            print(f"Model file not found for {name}. Skipping...")
            continue
        if not args.overwrite and os.path.exists(os.path.join(folder, sub, "test_results_aggregate.pkl")):
            print(f"Forecast results already exist for {name}. Skipping...")
            continue
        if not os.path.exists(os.path.join(folder, "config.json")):
            print(f"Config file not found for {name}. Skipping...")
            continue
        print(f"Imputing {name} ({i + 1})...")
        agg = run_folder(folder, data, args)
        if agg is not None:
            done[name] = agg
            print(f"Finished testing {name}: {agg}")
    return done


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Forecast by imputation with all the models in a directory.")
    parser.add_argument("--config_dir", type=str, required=True, help="Directory containing the saved model folders.")
    parser.add_argument("--test_npz", type=str, required=True, help="Test split (X_test, user_ids, months, cond_<tag> arrays).")
    parser.add_argument("--overwrite", type=utils.str2bool, default=True)
    parser.add_argument("--device", type=str, default="cuda:0")
    parser.add_argument("--num_imputation_samples", type=int, default=100)
    parser.add_argument("--batch_size", type=int, default=25000)
    parser.add_argument("--unobserved_window_length", type=int, default=24)
    parser.add_argument("--num_pseudo_gibbs_steps", type=int, default=1000)
    parser.add_argument("--num_metropolis_within_hasting_steps", type=int, default=0)
    main(parser.parse_args())
