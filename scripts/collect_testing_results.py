"""Test-set metrics for every trained run folder of a directory (CLI of the reference's
``scripts/collect_testing_results.py``: ``--config_dir --overwrite --device --num_rec_samples --batch_size
--batch_size_mc``), computed by the streamed device scoring instead of the ``[K,N,.]`` host buffers.

The reference rebuilds its data splits with ``preprocess_lib.prepare_data`` from raw CSV files; that pipeline is not
part of this package, so the test split comes from ``--test_npz``: an ``.npz`` with ``X_test [N,D]``, ``user_ids [N]``,
``months [N]`` and one ``cond_<tag> [N,.]`` array per condition tag of the run's ``conditioner.pkl``.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from behavio_temporal_vae_amd import results_lib, testing_lib, utils  # noqa: E402
from behavio_temporal_vae_amd.conditioning_lib import Conditioner  # noqa: E402
from behavio_temporal_vae_amd.vae_models import CVAE  # noqa: E402


def run_folder(folder, data, args):
    with open(os.path.join(folder, "config.json")) as f:
        config = json.load(f)
    conditioner = Conditioner.load(folder)
    X = np.asarray(data["X_test"], dtype=np.float32)
    condition_set = {tag: data["cond_" + tag] for tag in conditioner.tags}
    model = CVAE(input_dim=X.shape[1], conditioner=conditioner, **config["model"])
    model.load(folder)
    model.eval()
    metrics = testing_lib.mass_test_metrics(model, X, conditioner, condition_set, num_mc_samples=args.num_rec_samples,
                                            batch_size=args.batch_size, mc_sample_batch_size=args.batch_size_mc, device=args.device)
    frame = results_lib.test_results_frame(metrics, data["user_ids"], data["months"], "test")
    agg = results_lib.test_results_aggregate(metrics)
    results_lib.save_test_results(folder, {"test": frame}, {"test": agg})
    return agg


def main(args):
    data = dict(np.load(args.test_npz, allow_pickle=False))
    done = {}
    for i, name in enumerate(sorted(os.listdir(args.config_dir))):
        folder = os.path.join(args.config_dir, name)
        if not os.path.exists(os.path.join(folder, "trained_model.pt")):
            print(f"Model file not found for {name}. Skipping...")
            continue
        have = all(os.path.exists(os.path.join(folder, f)) for f in ("test_results_raw.csv", "test_results_aggregate.pkl"))
        if have and not args.overwrite:
            print(f"Test results already exist for {name}. Skipping...")
            continue
        if not os.path.exists(os.path.join(folder, "config.json")):
            print(f"Config file not found for {name}. Skipping...")
            continue
        print(f"Testing {name} ({i + 1})...")
        done[name] = run_folder(folder, data, args)
        print(f"Finished testing {name}: {done[name]}")
    return done


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Test all the models in a directory.")
    parser.add_argument("--config_dir", type=str, required=True, help="Directory containing the saved model folders.")
    parser.add_argument("--test_npz", type=str, required=True, help="Test split (X_test, user_ids, months, cond_<tag> arrays).")
    parser.add_argument("--overwrite", type=utils.str2bool, default=True)
    parser.add_argument("--device", type=str, default="cuda:0")
    parser.add_argument("--num_rec_samples", type=int, default=100)
    parser.add_argument("--batch_size", type=int, default=50000)
    parser.add_argument("--batch_size_mc", type=int, default=40)
    main(parser.parse_args())
