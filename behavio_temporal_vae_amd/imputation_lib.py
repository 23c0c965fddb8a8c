"""Forecast-by-imputation with pseudo-Gibbs sampling, API-compatible with ``src/imputation_lib.py``.

Each Gibbs sweep is: model forward with one latent draw -> decoder covariance (GEMM) -> per-profile
conditional Gaussian of the missing coordinates given the observed ones and one draw from it (one wave per
profile, ``btvae_cond_mvn_sample``) -> row-major masked write-back into ``x``.  Index and mask handling follows
the reference exactly: observed indices are read from row 0 of the mask only, ring-buffer slot is
``itx % num_samples``, the write-back consumes ``samples.flatten()`` in row-major mask order.

Two deliberate differences from the reference, both on code the reference's own scripts never reach with its shipped
configs: ``pseudo_gibbs`` stores every row's posterior parameters in ``z_rec["params"]`` (the reference writes row 0's
to all rows, src/imputation_lib.py:89), and the Normal-decoder conditional uses the unobserved coordinates' sigmas
(the reference indexes the observed ones, :37-38).  A profile with nothing missing raises ``ValueError``.
"""
from __future__ import annotations

import ctypes
import math
import time

import numpy as np
import torch

from . import _native as nv
from . import engine as E
from . import submodels


def timing_decorator(func):
    """Accumulates wall time in ``wrapper.total_time`` (reference: src/imputation_lib.py:6-14)."""
    def wrapper(*args, **kwargs):
        t0 = time.time()
        out = func(*args, **kwargs)
        wrapper.total_time += time.time() - t0
        return out
    wrapper.total_time = 0
    return wrapper


def _index_sets(D, observed_indices):
    obs = [int(i) for i in (observed_indices.tolist() if torch.is_tensor(observed_indices) else observed_indices)]
    un = [i for i in range(D) if i not in obs]
    return obs, un


class CondSplit:
    """The observed / unobserved partition of the D coordinates as the kernel wants it (bit mask of the observed ones,
    rank of every coordinate among the unobserved ones as a device array).  It is the same for every sweep of a chain,
    so the chains build it once (one host-to-device copy) instead of once per sweep."""

    def __init__(self, D, observed_indices, device):
        self.obs, self.un = _index_sets(D, observed_indices)
        self.D, self.U = D, len(self.un)
        if self.U == 0:
            raise ValueError("nothing to impute: every coordinate is observed")
        self.mask = 0
        for i in self.obs:
            self.mask |= 1 << i
        un_pos = torch.zeros(D, dtype=torch.int32)
        for r_, i in enumerate(self.un):
            un_pos[i] = r_
        self.un_pos = un_pos.to(device)
        self.obs_index = torch.tensor(self.obs, dtype=torch.long, device=device)


def _cond_rows(model, mu, sigma, x_full, eps, split):
    """Kernel call on flattened rows: returns (samples, mean, cov) with shapes [rows,U], [rows,U], [rows,U,U].
    ``x_full[rows, D]`` is read at the observed coordinates only."""
    D, U = model.output_dim, split.U
    rows = mu.shape[0]
    dev = mu.device
    nv.ensure_device(dev)
    tab = model.tables()
    Sp = E.dict_covariance_packed(tab, (sigma * sigma).contiguous())
    samp, mean = torch.empty(rows, U, device=dev), torch.empty(rows, U, device=dev)
    cov = torch.empty(rows, U, U, device=dev)
    if D <= 64:
        nv.call("btvae_cond_mvn_sample", nv.ptr(Sp), nv.ptr(mu), nv.ptr(x_full), nv.ptr(eps), split.mask, nv.ptr(split.un_pos), nv.ptr(samp),
                nv.ptr(mean), nv.ptr(cov), rows, D, U)
    else:                                             # the observed set as two 64-bit words
        words = (ctypes.c_ulonglong * 2)(split.mask & (2 ** 64 - 1), split.mask >> 64)
        nv.call("btvae_cond_mvn_sample_wide", nv.ptr(Sp), nv.ptr(mu), nv.ptr(x_full), nv.ptr(eps), ctypes.cast(words, ctypes.c_void_p),
                nv.ptr(split.un_pos), nv.ptr(samp), nv.ptr(mean), nv.ptr(cov), rows, D, U)
    return samp, mean, cov


def _conditional(model, param_dict, x_observed, observed_indices, eps=None, split=None, x_full=None):
    """``split`` / ``x_full`` (the whole ``[rows, D]`` profile tensor, read at the observed coordinates only) let a chain
    skip rebuilding the partition and the scattered copy of ``x_observed`` on every sweep."""
    D = model.output_dim
    mu = param_dict["mu"]
    if split is None:
        split = CondSplit(D, observed_indices, mu.device)
    obs, un = split.obs, split.un
    lead = mu.shape[:-1]
    mu_r = nv.f32(mu.reshape(-1, D))
    if isinstance(model, submodels.DictionaryGaussian):
        sg_r = nv.f32(param_dict["sigma"].reshape(-1, model.vocab_size))
        if x_full is None or x_full.shape != mu_r.shape or not x_full.is_contiguous() or x_full.dtype != torch.float32:
            if x_observed is None:                      # a chain handed its whole profile tensor in a form the kernel does not read
                if x_full is None:
                    raise ValueError("conditional parameters need the observed values: pass x_observed or x_full")
                x_observed = x_full.reshape(-1, D)[:, split.obs_index]
            xo = x_observed.expand(*lead, len(obs)).reshape(-1, len(obs))
            x_full = torch.zeros(mu_r.shape[0], D, device=mu.device)
            x_full[:, split.obs_index] = xo.float()
        if eps is None:
            eps = torch.zeros(mu_r.shape[0], len(un), device=mu.device)
        samp, mean, cov = _cond_rows(model, mu_r, sg_r, x_full, nv.f32(eps.reshape(-1, len(un))), split)
        U = len(un)
        return samp.view(*lead, U), mean.view(*lead, U), cov.view(*lead, U, U)
    if isinstance(model, submodels.GaussianNN):
        # Independent coordinates: conditioning leaves the unobserved marginals as they are.  The reference builds this
        # covariance from the sigmas of the OBSERVED coordinates (src/imputation_lib.py:37-38), which has the wrong
        # coordinates and, when the two sets differ in size, the wrong shape; that is not reproduced.
        mean = param_dict["mu"][..., un]
        sg = param_dict["sigma"][..., un]
        cov = torch.diag_embed(sg ** 2)
        samp = mean if eps is None else mean + sg * eps
        return samp, mean, cov
    raise NotImplementedError("This type of decoder is not implemented yet!")


@timing_decorator
@torch.no_grad()
def get_conditional_params(model, param_dict, x_observed, observed_indices):
    """Mean and covariance of the unobserved coordinates given the observed ones
    (reference: src/imputation_lib.py:16-42)."""
    _, mean, cov = _conditional(model, param_dict, x_observed, observed_indices)
    return mean, cov


@timing_decorator
@torch.no_grad()
def sample_conditional(model, param_dict, x_observed, observed_indices, num_samples=1, split=None, x_full=None):
    """One draw mean + chol(cov) eps plus the conditional parameters (reference: src/imputation_lib.py:44-57)."""
    D = model.output_dim
    U = split.U if split is not None else len(_index_sets(D, observed_indices)[1])
    lead = param_dict["mu"].shape[:-1]
    eps = torch.randn((*lead, U), device=param_dict["mu"].device)
    return _conditional(model, param_dict, x_observed, observed_indices, eps=eps, split=split, x_full=x_full)


def _conditions_tensor(conditioner, condition_set, device):
    c = torch.tensor(conditioner.transform(condition_set)).float().to(device)
    return c.squeeze(0) if c.ndim == 3 else c


def _scatter_index(missing_mask, n_expected):
    """Flat positions of the mask's True entries in row-major order: ``x[mask] = v.flatten()`` writes
    ``v.flatten()[t]`` to position ``idx[t]``."""
    idx = torch.nonzero(missing_mask.reshape(-1), as_tuple=False).reshape(-1)
    if idx.numel() != n_expected:
        raise RuntimeError(f"shape mismatch: value tensor of {n_expected} elements cannot be assigned to {idx.numel()} masked positions")
    return idx.contiguous()


def _is_random(conditioner):
    return "dir" in conditioner.types and conditioner.transformers[conditioner.tags[conditioner.types.index("dir")]].transform_style == "sample"


@timing_decorator
@torch.no_grad()
def pseudo_gibbs(model, x, missing_mask, conditioner, condition_set, num_samples=10, num_iter=100, verbose_freq=1000, device="cpu"):
    """Pseudo-Gibbs chain (reference: src/imputation_lib.py:59-98).  ``x`` is updated in place.  Returns the ring
    buffers (cond_x_samples, cond_x_mean, cond_x_cov, x_rec, z_rec) of the last ``num_samples`` sweeps."""
    observed_indices = torch.where((~missing_mask)[0])[0]     # row 0 only, as the reference does
    B, D, L = x.shape[0], model.input_dim, model.latent_dim
    dev = x.device
    split = CondSplit(D, observed_indices, dev)
    U = split.U
    dec = model.decoder
    # the reference probes shapes with one forward of row 0 and one conditional draw; consume the same RNG draws
    torch.randn((1, 1, L), device=dev); torch.randn((1, U), device=dev)
    Vw = dec.vocab_size if isinstance(dec, submodels.DictionaryGaussian) else D
    cond_x_samples = torch.zeros(num_samples, B, U, device=dev)
    cond_x_mean = torch.zeros(num_samples, B, U, device=dev)
    cond_x_cov = torch.zeros(num_samples, B, U, U, device=dev)
    x_rec = {"params": {"mu": torch.zeros(num_samples, B, D, device=dev), "sigma": torch.zeros(num_samples, B, Vw, device=dev)}}
    z_rec = {"params": {k: torch.zeros(num_samples, B, L, device=dev) for k in ("mu", "sigma")},
             "samples": torch.zeros(num_samples, B, L, device=dev)}
    conditions = _conditions_tensor(conditioner, condition_set, device)
    random_conditioning = _is_random(conditioner)
    idx = _scatter_index(missing_mask, B * U)
    for itx in range(num_iter):
        if itx % verbose_freq == 0:
            print(f"Iteration: {itx}")
        x_rec_, z_rec_ = model.forward(x, conditions)
        slot = itx % num_samples
        params0 = {k: v[0] for k, v in x_rec_["params"].items()}
        # x itself serves as the kernel's profile tensor: it is read at the observed coordinates only
        s, m, c = sample_conditional(dec, params0, None, observed_indices, split=split, x_full=x)
        cond_x_samples[slot], cond_x_mean[slot], cond_x_cov[slot] = s, m, c
        for k, v in x_rec_["params"].items():
            x_rec["params"][k][slot] = v[0]
        for k, v in z_rec_["params"].items():
            z_rec["params"][k][slot] = v
        z_rec["samples"][slot] = z_rec_["samples"][0]
        nv.call("btvae_masked_scatter", nv.ptr(x), nv.ptr(idx), nv.ptr(cond_x_samples[slot]), idx.numel())
        if random_conditioning:
            conditions = _conditions_tensor(conditioner, condition_set, device)
    return cond_x_samples, cond_x_mean, cond_x_cov, x_rec, z_rec


@timing_decorator
@torch.no_grad()
def metropolis_within_gibbs(model, x, missing_mask, conditioner, condition_set, num_samples=10, num_iter=100, verbose_freq=1000, device="cpu"):
    """Metropolis-within-Gibbs refinement of the latent state (reference: src/imputation_lib.py:100-156): propose
    z' ~ q(z|x,c), accept with min(1, w(z')/w(z)), w = p(x|z)p(z)/q(z|x); then redraw the missing coordinates."""
    observed_indices = torch.where((~missing_mask)[0])[0]
    B, D, L = x.shape[0], model.input_dim, model.latent_dim
    dev = x.device
    split = CondSplit(D, observed_indices, dev)
    U = split.U
    enc, dec = model.encoder, model.decoder
    conditions = _conditions_tensor(conditioner, condition_set, device)
    z = enc.sample(enc(torch.cat((x, conditions), dim=-1)))[0]
    torch.randn((1, U), device=dev)       # the reference's shape-probing conditional draw
    x_params = dec(torch.cat((z, conditions), dim=-1))
    cond_x_samples = torch.zeros(num_samples, B, U, device=dev)
    cond_x_mean = torch.zeros(num_samples, B, U, device=dev)
    cond_x_cov = torch.zeros(num_samples, B, U, U, device=dev)
    random_conditioning = _is_random(conditioner)
    idx = _scatter_index(missing_mask, B * U)
    prior = {k: v.to(dev) for k, v in model.prior_params.items()}
    accepted = 0
    for itx in range(num_iter):
        if itx % verbose_freq == 0:
            accepted = 0
            print(f"Iteration: {itx}")
        z_params = enc(torch.cat((x, conditions), dim=-1))
        x_params = dec(torch.cat((z, conditions), dim=-1))
        if random_conditioning:
            conditions = _conditions_tensor(conditioner, condition_set, device)
        z_prop = enc.sample(enc(torch.cat((x, conditions), dim=-1)))[0]
        x_params_prop = dec(torch.cat((z_prop, conditions), dim=-1))
        w_prop = dec.log_likelihood(x, param_dict=x_params_prop) + enc.log_likelihood(z_prop, param_dict=prior) - enc.log_likelihood(z_prop, param_dict=z_params)
        w_cur = dec.log_likelihood(x, param_dict=x_params) + enc.log_likelihood(z, param_dict=prior) - enc.log_likelihood(z, param_dict=z_params)
        rho = torch.minimum(torch.tensor([1.0]), torch.exp((w_prop - w_cur).cpu()))
        u = torch.rand((rho.shape[0],))
        acc = torch.where(u < rho)[0].to(dev)
        z[acc] = z_prop[acc]
        for k, v in x_params_prop.items():
            x_params[k][acc] = v[acc]
        accepted += acc.shape[0]
        slot = itx % num_samples
        s, m, c = sample_conditional(dec, x_params, None, observed_indices, split=split, x_full=x)
        cond_x_samples[slot], cond_x_mean[slot], cond_x_cov[slot] = s, m, c
        nv.call("btvae_masked_scatter", nv.ptr(x), nv.ptr(idx), nv.ptr(cond_x_samples[slot]), idx.numel())
        if (itx + 1) % verbose_freq == 0:
            print(f"Acceptance ratio: %{accepted / (verbose_freq * x.shape[0]) * 100}")
    return cond_x_samples, cond_x_mean, cond_x_cov


def _report(outer):
    if outer.total_time > 0:
        print(f"Percentage time spent in get_conditional_params: {get_conditional_params.total_time / outer.total_time * 100:.2f}%")
        print(f"Percentage time spent in sample_conditional: {sample_conditional.total_time / outer.total_time * 100:.2f}%")


@torch.no_grad()
def cvae_imputation(model, data, conditioner, condition_set, missing_data_init=None, num_samples=10, num_iter=100, warmup_steps=10,
                    verbose_freq=1000, device="cpu"):
    """Impute the NaN entries of ``data`` (reference: src/imputation_lib.py:158-189)."""
    model.to(device)
    model._prior_to(device)
    x = torch.tensor(data.copy()).float().to(device)
    missing_mask = torch.isnan(x)
    x[missing_mask] = x[~missing_mask].mean() if missing_data_init is None else missing_data_init
    x = x.contiguous()
    x_rec = z_rec = None
    if warmup_steps > 0:
        print("Pseudo-Gibbs is starting...")
        cs, cm, cc, x_rec, z_rec = pseudo_gibbs(model, x, missing_mask, conditioner, condition_set, num_samples=num_samples,
                                                num_iter=warmup_steps, verbose_freq=verbose_freq, device=device)
        if num_iter > 0:
            x[missing_mask] = cs[warmup_steps % num_samples].flatten()
        print("Pseudo-Gibbs has ended.")
        _report(pseudo_gibbs)
    else:
        cs, cm, cc = x, [None], [None]
    if num_iter > 0:
        print("Metropolis-within-Gibbs is starting...")
        cs, cm, cc = metropolis_within_gibbs(model, x, missing_mask, conditioner, condition_set, num_samples=num_samples,
                                             num_iter=num_iter, verbose_freq=verbose_freq, device=device)
        print("Metropolis-within-Gibbs has ended.")
        _report(metropolis_within_gibbs)
        x_rec, z_rec = None, None
    model.to("cpu")
    model._prior_to("cpu")
    return cs, cm, cc, x_rec, z_rec


def _batches(n, batch_size):
    return [(i, min(n, i + batch_size)) for i in range(0, n, batch_size)]


@torch.no_grad()
def mass_cvae_imputation(model, data, conditioner, condition_set, batch_size=1000, num_samples=10, num_iter=100, warmup_steps=10,
                         verbose_freq=1000, device="cpu"):
    """Batched imputation returning host arrays (reference: src/imputation_lib.py:191-214)."""
    U = int(np.isnan(data[0]).sum())
    N = data.shape[0]
    cs = np.zeros((num_samples, N, U), np.float32)
    cm = np.zeros((num_samples, N, U), np.float32)
    cc = np.zeros((num_samples, N, U, U), np.float32)
    bl = _batches(N, batch_size)
    for i, (b0, b1) in enumerate(bl):
        print(f"Batch {i+1}/{len(bl)}")
        s, m, c, _, _ = cvae_imputation(model, data[b0:b1], conditioner, {k: v[b0:b1] for k, v in condition_set.items()},
                                        num_samples=num_samples, num_iter=num_iter, warmup_steps=warmup_steps, verbose_freq=verbose_freq,
                                        device=device)
        cs[:, b0:b1], cm[:, b0:b1], cc[:, b0:b1] = s.cpu().numpy(), m.cpu().numpy(), c.cpu().numpy()
    return cs, cm, cc


def mixture_loglikelihood(cond_mean, cond_cov, target):
    """logsumexp_k log N(target; mean_k, cov_k) - log K on the device (reference: src/imputation_lib.py:230-231)."""
    K, B, U = cond_mean.shape
    dev = cond_mean.device
    nv.ensure_device(dev)
    packed = torch.empty(K * B, U * (U + 1) // 2, device=dev)
    nv.call("btvae_pack_lower", nv.ptr(cond_cov.contiguous()), nv.ptr(packed), K * B, U)
    logp = torch.empty(K * B, device=dev)
    nv.call("btvae_mvn_rows_fwd", nv.ptr(packed), nv.ptr(cond_mean.reshape(K * B, U).contiguous()), nv.ptr(nv.f32(target)), nv.ptr(logp), K * B, B, U)
    out = torch.empty(B, device=dev)
    nv.call("btvae_lse_over_k", nv.ptr(logp), nv.ptr(out), K, B)
    return out


@torch.no_grad()
def mass_cvae_imputation_with_loglikelihood(model, data, target_data, conditioner, condition_set, missing_data_init=None, batch_size=1000,
                                            num_samples=10, num_iter=100, warmup_steps=10, verbose_freq=1000, device="cpu"):
    """(loglikelihoods[N], cond_x_samples[num_samples,N,W]) (reference: src/imputation_lib.py:216-247)."""
    N = data.shape[0]
    lls = np.ndarray(shape=(N,), dtype=np.float32)
    samples = np.zeros((num_samples, N, target_data.shape[1]), np.float32)
    bl = _batches(N, batch_size)
    for i, (b0, b1) in enumerate(bl):
        print(f"Batch {i+1}/{len(bl)}")
        s, m, c, _, _ = cvae_imputation(model, data[b0:b1], conditioner, {k: v[b0:b1] for k, v in condition_set.items()},
                                        missing_data_init=missing_data_init, num_samples=num_samples, num_iter=num_iter,
                                        warmup_steps=warmup_steps, verbose_freq=verbose_freq, device=device)
        tgt = torch.tensor(target_data[b0:b1]).float().to(m.device)
        lls[b0:b1] = mixture_loglikelihood(m, c, tgt).cpu().numpy()
        samples[:, b0:b1] = s.cpu().numpy()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    return lls, samples
