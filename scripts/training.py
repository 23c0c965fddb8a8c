"""Training entry point with the reference's CLI (``scripts/training.py --config_path <json>``) and
``main(config)`` / ``train_model`` signatures, running the HIP path.

The config schema is the reference's (``data`` / ``model`` / ``train`` sections, ``save_dir``, ``save_tag``).
When the dataset named by ``config["data"]`` is not on disk, seeded synthetic profiles of the same shape are
used (``--synthetic-train/--synthetic-val`` sizes).  Conditions are encoded once and batches are assembled on
the device, instead of a per-item transform in DataLoader workers.
"""
import argparse
import datetime
import json
import os
import sys

import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd.vae_models import CVAE, make_writer          # noqa: E402
from behavio_temporal_vae_amd.synthetic import prepare_synthetic_data     # noqa: E402
from behavio_temporal_vae_amd.loaders import DeviceBatchLoader            # noqa: E402


def load_config(json_file):
    with open(json_file, "r") as f:
        return json.load(f)


def resolve_device(name):
    """The reference's configs name e.g. ``cuda:1``; fall back to cuda:0 when that index does not exist."""
    if isinstance(name, str) and name.startswith("cuda") and torch.cuda.is_available():
        idx = int(name.split(":")[1]) if ":" in name else 0
        return f"cuda:{idx if idx < torch.cuda.device_count() else 0}"
    return name


def train_model(model, trainset, valset, train_kwargs, writer=None):
    kw = dict(train_kwargs)
    kw["device"] = resolve_device(kw.get("device", "cuda:0"))
    trainloader = DeviceBatchLoader(trainset, batch_size=kw["batch_size"], shuffle=True, drop_last=True, device=kw["device"])
    valloader = DeviceBatchLoader(valset, batch_size=1024, shuffle=False, drop_last=False, device=kw["device"])
    model.fit(trainloader=trainloader, valloader=valloader, **kw, writer=writer)
    model.to("cpu")
    model.prior_params = {k: v.to("cpu") for k, v in model.prior_params.items()}
    model.eval()
    return model


def main(config, n_train=65536, n_val=8192):
    trainset, valset, conditioner = prepare_synthetic_data(config["data"], n_train, n_val)
    valset.num_samples = config["train"]["validation_mc_samples"]
    model = CVAE(input_dim=trainset.inputs.shape[-1], conditioner=conditioner, **config["model"])
    print("Number of encoder parameters:", model.encoder._num_parameters())
    print("Number of decoder parameters:", model.decoder._num_parameters())
    if config.get("save_dir") is not None:
        stamp = datetime.datetime.now().strftime("%b%d_%H-%M-%S")
        writer = make_writer(os.path.join(config["save_dir"], config["save_tag"] + stamp))
    else:
        writer = make_writer()
    with open(os.path.join(writer.log_dir, "config.json"), "w") as f:
        json.dump(config, f, indent=4)
    model = train_model(model, trainset, valset, config["train"], writer=writer)
    model.save()
    conditioner.save(model.log_dir)
    print(f"Model saved at {model.log_dir}")
    return model


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Train a conditional VAE with the configuration from a JSON file.")
    ap.add_argument("--config_path", type=str, default="./config_files/config0.json", help="Path to the JSON configuration file.")
    ap.add_argument("--synthetic-train", type=int, default=65536)
    ap.add_argument("--synthetic-val", type=int, default=8192)
    ap.add_argument("--epochs", type=int, default=None, help="override train.epochs")
    args = ap.parse_args()
    cfg = load_config(args.config_path)
    if args.epochs is not None:
        cfg["train"]["epochs"] = args.epochs
    main(cfg, args.synthetic_train, args.synthetic_val)
