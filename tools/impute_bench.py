#!/usr/bin/env python3
"""Timing of the pseudo-Gibbs imputation sweep at the reference's default batch (config0 shape: D=48, the last 24
coordinates missing, B=25,000 profiles, deterministic conditions): sweeps per second and profile-sweeps per second."""
import contextlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


class Cond:
    """Stand-in conditioner: passes a ready [B, C] array through (tags/types as a deterministic conditioner has them)."""
    tags, types, transformers, cond_dim = ["months"], ["circ"], {}, bench.C

    def transform(self, cs):
        return cs["c"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device("cuda:0")
    from behavio_temporal_vae_amd import imputation_lib as IL
    from behavio_temporal_vae_amd.vae_models import CVAE
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = CVAE(input_dim=bench.D, conditioner=bench.Cdim(bench.C), **bench.CONFIG0_MODEL).to(dev).eval()
    rng = np.random.default_rng(0)
    x, c = bench.synth_batch(rng, B)
    x = torch.from_numpy(x).to(dev)
    mask = torch.zeros(B, bench.D, dtype=torch.bool, device=dev)
    mask[:, 24:] = True
    x[mask] = 1.0
    cs = {"c": c}
    with contextlib.redirect_stdout(sys.stderr):
        IL.pseudo_gibbs(model, x, mask, Cond(), cs, num_samples=10, num_iter=3, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        IL.pseudo_gibbs(model, x, mask, Cond(), cs, num_samples=10, num_iter=iters, device=dev)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"B={B}: {dt * 1e3:.2f} ms per sweep, {B / dt / 1e6:.2f} M profile-sweeps/s")


if __name__ == "__main__":
    main()
