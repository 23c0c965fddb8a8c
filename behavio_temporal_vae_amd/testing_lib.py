"""Monte-Carlo scoring drivers with the reference's signatures (``src/testing_lib.py``).

``mass_loglikelihood`` streams: per batch the encoder runs once, each chunk of latent draws goes through the
decoder and the dictionary-Gaussian log-density kernels, and a running (max, sum) per profile folds the chunk
into the importance-weighted log-sum-exp on the device — no ``[K,N,.]`` decoder-parameter buffers exist.
``mass_reconstruction`` keeps the reference's return value (host tensors of every decoder parameter, latent
draw and log-likelihood) for callers that want them.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _native as nv
from . import engine as E
from .datasets import ConditionFeed
from .submodels import DictionaryGaussian


def _encode(model, x, c):
    """Eval/train-mode posterior parameters of a batch (reference: src/testing_lib.py:85)."""
    enc_in = torch.cat((x, c), dim=-1).contiguous()
    post, _ = E.net_forward(model.encoder.plan(), enc_in, model.training, None, model.encoder.head_specs())
    return post


def _decode_chunk(model, post, c, eps, x, B):
    """Decoder parameters and log p(x|z,c) for ``eps[kc,B,L]`` draws (reference: src/testing_lib.py:91-97)."""
    kc, L = eps.shape[0], model.latent_dim
    zc, _, _ = E.reparam_kl(post["mu"], post["sigma"], eps.reshape(-1, L), c, kc, False, B)
    dec = model.decoder
    is_dict = isinstance(dec, DictionaryGaussian)
    lik, sq = E.net_forward(dec.plan(), zc, model.training, None, dec.head_specs(), want_sq=("sigma",) if is_dict else ())
    if is_dict:
        logp = E.dict_loglik_fwd(dec.tables(), x, lik["mu"], sq["sigma"])
    else:
        if "sigma" not in lik:
            lik["sigma"] = torch.full_like(lik["mu"], float(dec.sigma_fixed))
        logp = E.diag_loglik(x, lik["mu"], lik["sigma"])[0]
    return zc, lik, logp.view(kc, B)


def _chunks(K, step):
    return [(i, min(K, i + step)) for i in range(0, K, step)]


@torch.no_grad()
def iw_loglikelihood_device(model, x, c, num_mc_samples, mc_sample_batch_size=100, eps=None, return_rlls=False, k_shard=None):
    """Importance-weighted log-likelihood of device-resident profiles ``x[N,D]`` with conditions ``c[N,C]``:
    ll[n] = -log K + logsumexp_k(-log q(z_k) + log p(x|z_k) + log p(z_k)) (reference: src/testing_lib.py:10-14).
    Everything stays on the device; ``eps[K,N,L]`` may be injected (tests), else it is drawn per chunk.

    ``k_shard=(rank, world[, group])`` splits the K draws over the ranks of a process group instead of the profiles
    (every rank passes the same ``x``): rank r runs draws ``shard_range(K, r, world)`` and the per-profile running
    (max, sum) states are merged by ``parallel.combine_lse`` -- two all-reduces of N floats -- so every rank returns the
    full-K ``ll``.  ``rlls`` then holds this rank's draws only (rows ``k0:k1``, the rest zero)."""
    model._check_arena()
    N, L, K = x.shape[0], model.latent_dim, num_mc_samples
    x, c = nv.f32(x), nv.f32(c)
    post = _encode(model, x, c)
    m_run = torch.full((N,), -math.inf, device=x.device)
    s_run = torch.zeros(N, device=x.device)
    rlls = (torch.zeros(K, N, device=x.device) if k_shard is not None else torch.empty(K, N, device=x.device)) if return_rlls else None
    k0, k1, group = 0, K, None
    if k_shard is not None:
        from .loaders import shard_range
        k0, k1 = shard_range(K, int(k_shard[0]), int(k_shard[1]))
        group = k_shard[2] if len(k_shard) > 2 else None
    for j0, j1 in _chunks(k1 - k0, mc_sample_batch_size):
        i0, i1 = k0 + j0, k0 + j1
        e = eps[i0:i1].contiguous() if eps is not None else torch.randn((i1 - i0, N, L), device=x.device)
        _, _, logp = _decode_chunk(model, post, c, e, x, N)
        nv.call("btvae_iw_lse_update", nv.ptr(logp), nv.ptr(e.reshape(-1, L)), nv.ptr(post["mu"]), nv.ptr(post["sigma"]),
                nv.ptr(m_run), nv.ptr(s_run), i1 - i0, N, L, int(j0 == 0))
        if rlls is not None:
            rlls[i0:i1] = logp
    if k_shard is not None:
        from .parallel import combine_lse
        combine_lse(m_run, s_run, group)
    ll = torch.empty(N, device=x.device)
    nv.call("btvae_iw_lse_finalize", nv.ptr(m_run), nv.ptr(s_run), nv.ptr(ll), N, K)
    # per-profile KL to the standard-normal prior (reference: src/testing_lib.py:12)
    _, kl, _ = E.reparam_kl(post["mu"], post["sigma"], torch.zeros(N, L, device=x.device), None, 1, False, N)
    return ll, rlls, kl


@torch.no_grad()
def test_metrics_device(model, x, c, num_mc_samples, mc_sample_batch_size=40, eps=None, eps_x=None):
    """What ``scripts/collect_testing_results.py:84-113`` derives from ``mass_reconstruction``'s host buffers, streamed on
    the device: importance-weighted log-likelihood, KL, and the l1 / l2 errors of one posterior-predictive sample per
    latent draw -- err_p[n] = (mean_k sum_d |x_s[k,n,d] - x[n,d]|^p)^(1/p).  ``eps[K,N,L]`` (latent noise) and
    ``eps_x[K,N,V or D]`` (sample noise) may be injected."""
    model._check_arena()
    N, L, K, D = x.shape[0], model.latent_dim, num_mc_samples, model.input_dim
    x, c = nv.f32(x), nv.f32(c)
    dev = x.device
    dec = model.decoder
    is_dict = isinstance(dec, DictionaryGaussian)
    post = _encode(model, x, c)
    m_run, s_run = torch.empty(N, device=dev), torch.empty(N, device=dev)
    e1, e2 = torch.empty(N, device=dev), torch.empty(N, device=dev)
    Uh = dec.get_SigmaMapper() if is_dict else None
    for i0, i1 in _chunks(K, mc_sample_batch_size):
        kc = i1 - i0
        e = eps[i0:i1].contiguous() if eps is not None else torch.randn((kc, N, L), device=dev)
        _, lik, logp = _decode_chunk(model, post, c, e, x, N)
        nv.call("btvae_iw_lse_update", nv.ptr(logp), nv.ptr(e.reshape(-1, L)), nv.ptr(post["mu"]), nv.ptr(post["sigma"]),
                nv.ptr(m_run), nv.ptr(s_run), kc, N, L, int(i0 == 0))
        W = lik["sigma"].shape[1]
        ex = eps_x[i0:i1].reshape(-1, W).contiguous() if eps_x is not None else torch.randn((kc * N, W), device=dev)
        xs = torch.empty(kc * N, D, device=dev)
        if is_dict:          # mu + Uhat (eps * sigma)   (src/submodels.py:160-164)
            scaled = (ex * lik["sigma"]).contiguous()
            nv.call("btvae_matmul_nt_add", nv.ptr(scaled), W, nv.ptr(Uh), W, nv.ptr(lik["mu"]), D, nv.ptr(xs), D, kc * N, D, W)
        else:
            torch.addcmul(lik["mu"], lik["sigma"], ex, out=xs)
        nv.call("btvae_sample_error_update", nv.ptr(xs), nv.ptr(x), nv.ptr(e1), nv.ptr(e2), kc, N, D, int(i0 == 0))
    ll = torch.empty(N, device=dev)
    nv.call("btvae_iw_lse_finalize", nv.ptr(m_run), nv.ptr(s_run), nv.ptr(ll), N, K)
    _, kl, _ = E.reparam_kl(post["mu"], post["sigma"], torch.zeros(N, L, device=dev), None, 1, False, N)
    return {"loglikelihood": ll, "kl": kl, "l_1": e1 / K, "l_2": torch.sqrt(e2 / K)}


@torch.no_grad()
def mass_test_metrics(model, x, conditioner, condition_set, num_mc_samples=100, batch_size=50000, mc_sample_batch_size=40, device="cpu"):
    """Per-profile test metrics of a whole set as numpy arrays {"loglikelihood", "kl", "l_1", "l_2"} without the
    ``[K,N,.]`` host buffers (27 GB at K=100 on the reference's test split) that the reference's collection script
    goes through.  Deterministic conditioners only (the sampled-condition case keeps the buffer route)."""
    if conditioner.is_random() if hasattr(conditioner, "is_random") else False:
        raise NotImplementedError("streamed test metrics need deterministic conditions; use mass_reconstruction for sampled ones")
    N = x.shape[0]
    model.to(device)
    model._prior_to(device)
    out = {k: np.empty(N, np.float32) for k in ("loglikelihood", "kl", "l_1", "l_2")}
    feed = ConditionFeed(x, condition_set, conditioner, batch_size, device=device)
    for k, (b0, b1, xb, cb) in enumerate(feed):
        print(f"Batch {k+1}/{len(feed)}")
        res = test_metrics_device(model, xb, cb, num_mc_samples, min(mc_sample_batch_size, num_mc_samples))
        for key, v in res.items():
            out[key][b0:b1] = v.cpu().numpy()
    model.to("cpu")
    model._prior_to("cpu")
    return out


test_metrics_device.__test__ = False      # not a pytest test despite the name


@torch.no_grad()
def mass_loglikelihood(model, x, conditioner, condition_set, num_mc_samples=1, batch_size=10000, mc_sample_batch_size=100, device="cpu"):
    """(loglikelihood[N], rlls[K,N], kl[N]) as numpy arrays (reference: src/testing_lib.py:6-16)."""
    if conditioner.is_random() if hasattr(conditioner, "is_random") else False:
        _, z_rec, rlls = mass_reconstruction(model, x, conditioner, condition_set, num_mc_samples=num_mc_samples, batch_size=batch_size,
                                             mc_sample_batch_size=mc_sample_batch_size, device=device)
        return _loglik_from_buffers(model, z_rec, rlls)
    if num_mc_samples <= mc_sample_batch_size:
        mc_sample_batch_size = num_mc_samples
        print(f"Setting the batch size for MC samples to {mc_sample_batch_size}.")
    N = x.shape[0]
    model.to(device)
    model._prior_to(device)
    ll, kl = np.empty(N, np.float32), np.empty(N, np.float32)
    rlls = np.empty((num_mc_samples, N), np.float32)
    feed = ConditionFeed(x, condition_set, conditioner, batch_size, device=device)
    for k, (b0, b1, xb, cb) in enumerate(feed):
        print(f"Batch {k+1}/{len(feed)}")
        l_, r_, k_ = iw_loglikelihood_device(model, xb, cb, num_mc_samples, mc_sample_batch_size, return_rlls=True)
        ll[b0:b1], rlls[:, b0:b1], kl[b0:b1] = l_.cpu().numpy(), r_.cpu().numpy(), k_.cpu().numpy()
    model.to("cpu")
    model._prior_to("cpu")
    return ll, rlls, kl


def _loglik_from_buffers(model, z_rec, rlls):
    """The reference's host-side combination for the sampled-condition case (src/testing_lib.py:10-14)."""
    from .utils import kl_divergence, log_prob
    post = log_prob("Normal", z_rec["params"], z_rec["samples"]).sum(-1)
    prior = log_prob("Normal", model.prior_params, z_rec["samples"]).sum(-1)
    kl = kl_divergence("Normal", z_rec["params"], model.prior_params).sum(-1)
    ll = -math.log(rlls.shape[0]) + torch.logsumexp(-post + rlls + prior, dim=0)
    return ll.numpy(), rlls.numpy(), kl.numpy()


@torch.no_grad()
def mass_reconstruction(model, x, conditioner, condition_set, num_mc_samples=1, batch_size=10000, mc_sample_batch_size=100, device="cpu"):
    """Decoder parameters ``x_rec``, posterior parameters/draws ``z_rec`` and log-likelihoods ``rlls[K,N]`` of every
    profile and draw, as host tensors (reference: src/testing_lib.py:19-131)."""
    if num_mc_samples <= mc_sample_batch_size:
        mc_sample_batch_size = num_mc_samples
        print(f"Setting the batch size for MC samples to {mc_sample_batch_size}.")
    N, K, L, D = x.shape[0], num_mc_samples, model.latent_dim, model.input_dim
    random_conditioning = conditioner.is_random() if hasattr(conditioner, "is_random") else False
    dec = model.decoder
    widths = {"mu": D, "sigma": dec.vocab_size if isinstance(dec, DictionaryGaussian) else D}
    x_rec = {"params": {k: torch.zeros(K, N, w) for k, w in widths.items()}}
    if random_conditioning:
        z_rec = {"params": {k: torch.zeros(K, N, L) for k in ("mu", "sigma")}}
    else:
        z_rec = {"params": {k: torch.zeros(N, L) for k in ("mu", "sigma")}}
    z_rec["samples"] = torch.zeros(K, N, L)
    rlls = torch.zeros(K, N)

    model.to(device)
    model._prior_to(device)
    model._check_arena()
    feed = ConditionFeed(x, condition_set, conditioner, batch_size, num_samples=K, device=device)
    for k, (b0, b1, xb, cb) in enumerate(feed):
        print(f"Batch {k+1}/{len(feed)}")
        nb = b1 - b0
        xb = nv.f32(xb)
        if random_conditioning:
            for i0, i1 in _chunks(K, mc_sample_batch_size):
                x_dict, z_dict = model.reconstruct(inputs=xb, conditions=cb[:, i0:i1, :].contiguous(), num_mc_samples=i1 - i0)
                for p in widths:
                    x_rec["params"][p][i0:i1, b0:b1] = x_dict["params"][p].cpu()
                for p in ("mu", "sigma"):
                    z_rec["params"][p][i0:i1, b0:b1] = z_dict["params"][p].cpu()
                z_rec["samples"][i0:i1, b0:b1] = z_dict["samples"].cpu()
                rlls[i0:i1, b0:b1] = dec.log_likelihood(xb, x_dict["params"]).cpu()
        else:
            cb = nv.f32(cb)
            post = _encode(model, xb, cb)
            for p in ("mu", "sigma"):
                z_rec["params"][p][b0:b1] = post[p].cpu()
            for i0, i1 in _chunks(K, mc_sample_batch_size):
                eps = torch.randn((i1 - i0, nb, L), device=xb.device)
                zc, lik, logp = _decode_chunk(model, post, cb, eps, xb, nb)
                z_rec["samples"][i0:i1, b0:b1] = zc.view(i1 - i0, nb, -1)[..., :L].cpu()
                for p in widths:
                    x_rec["params"][p][i0:i1, b0:b1] = lik[p].view(i1 - i0, nb, -1).cpu()
                rlls[i0:i1, b0:b1] = logp.cpu()
        if torch.cuda.is_available() and device != "cpu":
            torch.cuda.empty_cache()
    model.to("cpu")
    model._prior_to("cpu")
    return x_rec, z_rec, rlls
