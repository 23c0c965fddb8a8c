#!/usr/bin/env python3
"""One-GPU check of the RCCL call sequence: a one-rank "nccl" process group with every collective of the data-parallel
training step issued (``enable_data_parallel(force_collectives=True)``: each is an identity at world size 1).

* gradient sync only: four training steps must equal the plain single-device steps BIT FOR BIT -- a span reduced before
  the kernels that write it have finished, or read by the optimiser before its reduction has, would change bits;
* with ``sync_batchnorm=True`` (all_gather_into_tensor of the statistics, sum all-reduce of the backward sums) the same
  steps must agree to 1e-5: that path merges the statistics with other kernels than the single-device one;
* the scoring collectives (``reduce_loglik``, ``combine_lse``, ``all_gather_rows``) return their inputs.

Prints ``rccl smoke ok`` and exits 0; any mismatch raises.  Says nothing about scaling."""
import contextlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Cdim:
    def __init__(self, c):
        self.cond_dim = c


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from conftest import golden_state, load_golden
    from behavio_temporal_vae_amd.vae_models import CVAE
    from behavio_temporal_vae_amd import parallel as PL

    g = load_golden("tiny_dict_2d")
    T = lambda a: torch.from_numpy(np.array(a)).to(dev)
    steps = g["steps_x"].shape[0]

    def run(mode):
        with contextlib.redirect_stdout(sys.stderr):
            m = CVAE(input_dim=int(g["D"]), conditioner=Cdim(int(g["C"])), **json.loads(str(g["model_cfg"])))
        m.load_state_dict(golden_state(g))
        m = m.to(dev).train()
        m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
        if mode != "single":
            m.enable_data_parallel(sync_batchnorm=(mode == "sync"), force_collectives=True)
            assert m._dp_active and (m._bn_sync is not None) == (mode == "sync")
        opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
        losses = []
        for t in range(steps):
            l = m._train_step(T(g["steps_x"][t]), T(g["steps_c"][t]), opt, eps=T(g["steps_eps"][t]))
            losses.append(torch.stack([l[k] for k in ("loss", "elbo", "rll", "kl")]).clone())
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if v.is_floating_point()}

    l0, s0 = run("single")
    l1, s1 = run("grads")
    assert torch.equal(l0, l1), (l0, l1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    l2, s2 = run("sync")
    assert torch.allclose(l0, l2, rtol=1e-5, atol=1e-6), (l0, l2)
    for k in s0:
        assert torch.allclose(s0[k], s2[k], rtol=1e-4, atol=2e-6), (k, (s0[k] - s2[k]).abs().max())

    tot, cnt = PL.reduce_loglik(1.5, 3, device=dev)
    assert (tot, cnt) == (1.5, 3)
    loc = torch.arange(6, dtype=torch.float32, device=dev).reshape(2, 3)
    assert torch.equal(PL.all_gather_rows(loc)[0], loc)
    mr, sr = torch.tensor([0.5, -1.0], device=dev), torch.tensor([2.0, 3.0], device=dev)
    m2, s2_ = PL.combine_lse(mr.clone(), sr.clone())
    assert torch.equal(m2, mr) and torch.equal(s2_, sr)
    dist.barrier()
    torch.cuda.synchronize()
    print("rccl smoke ok: grad-sync steps bit-identical to single-device; sync-BN steps within 1e-5; losses", l1[-1].tolist())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
