"""GPU: data-parallel training with BatchNorm statistics spanning the ranks.  Two processes share the one GPU of the
test box and talk over gloo (device tensors staged through the host by ``parallel.py``; on a multi-GPU node the same
code runs over RCCL).  With synchronised BatchNorm, two ranks on half a batch each must reproduce the single-device
step on the whole batch: loss terms per step and the parameters after four optimiser steps."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_state, load_golden

pytestmark = pytest.mark.gpu


class Cdim:
    def __init__(self, c):
        self.cond_dim = c


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(g, dev):
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = json.loads(str(g["model_cfg"]))
    m = CVAE(input_dim=int(g["D"]), conditioner=Cdim(int(g["C"])), **cfg)
    m.load_state_dict(golden_state(g))
    m = m.to(dev).train()
    m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    return m


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        T = lambda a: torch.from_numpy(np.array(a)).to(dev)
        g = load_golden("tiny_dict_2d")
        steps = g["steps_x"].shape[0]
        B = g["steps_x"].shape[1]
        lo, hi = rank * B // world, (rank + 1) * B // world
        out = {}
        for mode in ("single", "sync", "per_rank", "per_rank_module_idiom"):
            m = _build(g, dev)
            if mode != "single":
                m.enable_data_parallel(sync_batchnorm=(mode == "sync"))
            opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
            losses = []
            for t in range(steps):
                x, c, eps = T(g["steps_x"][t]), T(g["steps_c"][t]), T(g["steps_eps"][t])
                if mode != "single":
                    x, c, eps = x[lo:hi].contiguous(), c[lo:hi].contiguous(), eps[:, lo:hi].contiguous()
                if mode == "per_rank_module_idiom":
                    # the reference's idiom on the module-level API: autograd fills the gradient arena, the caller averages it
                    from test_model_gpu import _FixedRandn
                    m.prior_params = {k: v.to(dev) for k, v in m.prior_params.items()}
                    with _FixedRandn(eps):
                        xd, zd = m(x, c, num_mc_samples=eps.shape[0])
                    opt.zero_grad(set_to_none=True)
                    l = m.loss(x, xd["params"], zd["params"], beta=0.7, prior_params=m.prior_params)
                    l["loss"].backward()
                    m.sync_gradients()
                    opt.step(max_norm=1.0)
                    losses.append([l[k].item() for k in ("loss", "elbo", "rll", "kl")])
                    continue
                l = m._train_step(x, c, opt, eps=eps)
                losses.append([l[k].item() for k in ("loss", "elbo", "rll", "kl")])
            sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if v.is_floating_point()}
            out[mode] = (np.array(losses), sd)
            from behavio_temporal_vae_amd import engine as E
            E.BN_SYNC = None
        # the same with a hidden width that is not a multiple of 32 (activation and gradient rows padded to a pitch of 96):
        # the statistics-spanning BatchNorm kernels on padded matrices, residual units included
        from behavio_temporal_vae_amd.vae_models import CVAE as _CVAE
        from behavio_temporal_vae_amd import engine as E
        assert E._rows_on_lines(2, 72, dev).stride(0) == 96
        cfgp = {"latent_dim": 6, "distribution_dict": {
            "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": 72,
                          "num_hidden_layers": 2, "dropout": False, "batch_normalization": True},
            "likelihood": {"dist_type": "dict-gauss", "vocab_size": 30, "sigma_lim": 1e-4, "marginal_std_lim": 0.1, "mu_lower_lim": -3,
                           "num_neurons": 72, "num_hidden_layers": 2, "dropout": False, "batch_normalization": True}}}
        gen = torch.Generator().manual_seed(77)
        Bp, Sp_, Dp, Cp = 96, 3, 12, 4
        xp = torch.randn(4, Bp, Dp, generator=gen).clamp(-3, 5).to(dev)
        cp = torch.randn(4, Bp, Cp, generator=gen).to(dev)
        ep = torch.randn(4, Sp_, Bp, 6, generator=gen).to(dev)
        lo, hi = rank * Bp // world, (rank + 1) * Bp // world
        for mode in ("single", "sync"):
            torch.manual_seed(5)
            m = _CVAE(input_dim=Dp, conditioner=Cdim(Cp), **cfgp).to(dev).train()
            m.train_kwargs = {"num_mc_samples": Sp_, "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
            if mode == "sync":
                m.enable_data_parallel(sync_batchnorm=True)
            opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
            losses = []
            for t in range(4):
                x, c, eps = xp[t], cp[t], ep[t]
                if mode == "sync":
                    x, c, eps = x[lo:hi].contiguous(), c[lo:hi].contiguous(), eps[:, lo:hi].contiguous()
                l = m._train_step(x, c, opt, eps=eps)
                losses.append([l[k].item() for k in ("loss", "elbo", "rll", "kl")])
            sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if v.is_floating_point()}
            out["padded_" + mode] = (np.array(losses), sd)
            E.BN_SYNC = None
        # scoring with the Monte-Carlo draws split over the ranks (same profiles on both), ragged chunks
        from behavio_temporal_vae_amd import testing_lib
        gs = load_golden("scoring_tiny")
        cfg = json.loads(str(gs["model_cfg"]))
        from behavio_temporal_vae_amd.vae_models import CVAE
        ms = CVAE(input_dim=int(gs["D"]), conditioner=Cdim(4), **cfg)
        ms.load_state_dict(golden_state(gs))
        ms = ms.to(dev).eval()
        xs, cs, es = T(gs["x"]), T(gs["c"]), T(gs["eps"])
        K = int(gs["K"])
        ll_full, _, _ = testing_lib.iw_loglikelihood_device(ms, xs, cs, K, mc_sample_batch_size=3, eps=es)
        ll_shard, rl_shard, _ = testing_lib.iw_loglikelihood_device(ms, xs, cs, K, mc_sample_batch_size=2, eps=es, return_rlls=True,
                                                                    k_shard=(rank, world))
        out["score"] = (ll_full.cpu().numpy(), ll_shard.cpu().numpy(), rl_shard.cpu().numpy())
        q.put((rank, out, None))
        dist.destroy_process_group()
    except Exception as exc:     # surface the failure in the parent instead of a silent non-zero exit
        import traceback
        q.put((rank, None, traceback.format_exc()))
        raise exc


def test_syncbn_two_ranks_equal_single_device():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=170) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, out, err in res:
        assert err is None, err
    g = load_golden("tiny_dict_2d")
    for rank, out, _ in res:
        single_l, single_sd = out["single"]
        sync_l, sync_sd = out["sync"]
        per_l, per_sd = out["per_rank"]
        # the module-level idiom with sync_gradients() lands on the weights of the fused data-parallel step
        _, idiom_sd = out["per_rank_module_idiom"]
        for k, v in per_sd.items():
            if "running_" not in k:               # BatchNorm running statistics are per rank in both runs
                assert np.abs(idiom_sd[k] - v).max() <= 5e-4 * (np.abs(v).max() + 1e-6), k
        # the single-device run is itself pinned to the reference
        assert np.max(np.abs(single_l - g["steps_losses"]) / (np.abs(g["steps_losses"]) + 1e-6)) < 1e-4
        assert np.max(np.abs(sync_l - single_l) / (np.abs(single_l) + 1e-6)) < 1e-4, (rank, sync_l, single_l)
        for k, v in single_sd.items():
            assert np.abs(sync_sd[k] - v).max() <= 5e-4 * (np.abs(v).max() + 1e-6), k
        # per-rank statistics (the default) are a different computation: the check above is not vacuous
        assert np.max(np.abs(per_l - single_l) / (np.abs(single_l) + 1e-6)) > 1e-3
        pl, psd = out["padded_single"]
        ql, qsd = out["padded_sync"]
        assert np.max(np.abs(ql - pl) / (np.abs(pl) + 1e-6)) < 1e-4, (rank, ql, pl)
        for k, v in psd.items():
            assert np.abs(qsd[k] - v).max() <= 5e-4 * (np.abs(v).max() + 1e-6), k
    assert np.array_equal(res[0][1]["sync"][0], res[1][1]["sync"][0])        # both ranks report the same averaged terms
    for k, v in res[0][1]["per_rank_module_idiom"][1].items():               # averaged gradients: both ranks hold the same weights
        if "running_" not in k:
            assert np.array_equal(v, res[1][1]["per_rank_module_idiom"][1][k]), k
    # K-sharded scoring == unsharded == the reference's golden ll on every rank; the ranks' draw ranges tile K
    gs = load_golden("scoring_tiny")
    rl = np.zeros_like(gs["rlls"])
    for rank, out, _ in res:
        ll_full, ll_shard, rl_shard = out["score"]
        assert np.max(np.abs(ll_full - gs["ll"]) / (np.abs(gs["ll"]) + 1e-6)) < 1e-4
        assert np.max(np.abs(ll_shard - gs["ll"]) / (np.abs(gs["ll"]) + 1e-6)) < 1e-4
        assert np.max(np.abs(ll_shard - ll_full) / (np.abs(ll_full) + 1e-6)) < 1e-5
        rl += rl_shard
    assert np.max(np.abs(rl - gs["rlls"]) / (np.abs(gs["rlls"]) + 1e-6)) < 1e-4
