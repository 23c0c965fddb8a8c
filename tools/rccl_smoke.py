#!/usr/bin/env python3
"""One-GPU smoke of the RCCL code path: a single-rank "nccl" process group, the gradient sync forced on (its
all-reduces are identities at world size 1), a few training steps, then the sharded-scoring reduce.  Checks that the
collectives enqueue and complete next to the library's kernels on the current stream; says nothing about scaling."""
import contextlib
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from behavio_temporal_vae_amd.vae_models import CVAE
    from behavio_temporal_vae_amd.parallel import reduce_loglik
    torch.manual_seed(1)
    with contextlib.redirect_stdout(sys.stderr):
        model = CVAE(input_dim=bench.D, conditioner=bench.Cdim(bench.C), **bench.CONFIG0_MODEL)
    model = model.to(dev).train()
    model.train_kwargs = dict(bench.CONFIG0_TRAIN)
    model.enable_data_parallel()
    model._dp_world = model._dp_sync.world = 2          # force the sync path; sums over one rank, then halves
    optim = model.get_optimizer(lr=1e-3, weight_decay=1e-5)
    import numpy as np
    rng = np.random.default_rng(0)
    x, c = bench.synth_batch(rng, 1024)
    x, c = torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)
    for i in range(3):
        loss = model.train_core((x, c), optim)
    dist.barrier()
    torch.cuda.synchronize()
    vals = {k: float(v.item()) for k, v in loss.items()}
    tot, cnt = reduce_loglik(1.5, 3, device=dev)
    assert all(v == v for v in vals.values()), vals
    assert (tot, cnt) == (1.5, 3)
    print("rccl smoke ok", vals)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
