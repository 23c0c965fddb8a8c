"""Distribution networks of the conditional VAE, API-compatible with the reference's ``src/submodels.py``.

The ``torch.nn`` modules here are parameter containers only (same attribute names and Sequential
indices as the reference, so checkpoints load unchanged: ``...input_layer.0.weight``, ``...2.running_mean``).
All arithmetic runs in the HIP library through :mod:`engine`; tensors must live on an MI355X.
Outputs of these module-level calls carry no autograd graph: training goes through
``CVAE.train_core`` / ``CVAE.loss_and_backward`` which run the hand-written backward kernels.
"""
from __future__ import annotations

import math

import torch

from . import _native as nv
from . import engine as E
from .autograd_ops import apply_net, dict_log_likelihood
from .utils import kl_divergence, log_prob, matrix_normalizer  # noqa: F401  (re-exported like the reference does)

ACTIVATION = torch.nn.ReLU()
_INF = float("inf")


def _stack(n_in, n_out, relu, bn, dropout, rate):
    mods = [torch.nn.Linear(n_in, n_out)]
    if relu:
        mods.append(ACTIVATION)
    return mods, bn, dropout, rate


class NNBlock(torch.nn.Module):
    """Input Linear+ReLU, ``num_hidden_layers`` residual Linear+ReLU units, output Linear; BatchNorm and
    Dropout are appended to the ReLU units when enabled (reference: src/submodels.py:6-35).  Construction
    order matches the reference so that a given torch seed produces the same initial weights."""

    def __init__(self, input_dim, output_dim, num_neurons=50, num_hidden_layers=2, dropout=False, dropout_rate=0.5,
                 batch_normalization=False, resnet=True, **_):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.num_neurons, self.num_layers, self.resnet = num_neurons, num_hidden_layers, resnet
        S, L = torch.nn.Sequential, torch.nn.Linear
        self.input_layer = S(L(input_dim, num_neurons), ACTIVATION)
        self.middle_layers = torch.nn.ModuleList(S(L(num_neurons, num_neurons), ACTIVATION) for _ in range(num_hidden_layers))
        self.output_layer = S(L(num_neurons, output_dim))
        relu_units = [self.input_layer, *self.middle_layers]
        if batch_normalization:
            for u in relu_units:
                u.append(torch.nn.BatchNorm1d(num_neurons))
        if dropout:
            for u in relu_units:
                u.append(torch.nn.Dropout(dropout_rate))

    def _num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class ParameterizerNN(torch.nn.Module):
    """Shared trunk + one two-layer head per distribution parameter (reference: src/submodels.py:41-69)."""

    def __init__(self, input_dim, output_dim, dist_params=["mu"], num_hidden_layers=2, num_neurons=50, dropout=False,
                 dropout_rate=0.5, batch_normalization=False, **_):
        super().__init__()
        self.dist_params = dist_params
        self.block_dict = torch.nn.ModuleDict()
        common = dict(num_neurons=num_neurons, dropout=dropout, dropout_rate=dropout_rate, batch_normalization=batch_normalization)
        trunk = NNBlock(input_dim, num_neurons, num_hidden_layers=num_hidden_layers, **common)
        trunk.output_layer.append(ACTIVATION)
        if batch_normalization:
            trunk.output_layer.append(torch.nn.BatchNorm1d(num_neurons))
        if dropout:
            trunk.output_layer.append(torch.nn.Dropout(dropout_rate))
        self.block_dict["input"] = trunk
        if type(output_dim) is int:
            widths = {p: output_dim for p in dist_params}
        elif type(output_dim) is list:
            widths = {p: w for w, p in zip(output_dim, dist_params)}
        elif type(output_dim) is dict:
            widths = dict(output_dim)
        else:
            widths = {}
        for name, w in widths.items():
            self.block_dict[name] = NNBlock(num_neurons, w, num_hidden_layers=0, **common)
        self._plan = None

    def plan(self) -> E.NetPlan:
        if self._plan is None:
            holder = type("H", (), {})()
            holder.parameterizer = self
            object.__setattr__(self, "_plan", E.NetPlan(holder))
        return self._plan

    def forward(self, inputs):
        """Raw (un-clamped) head outputs as a dict, rows = ``inputs`` flattened to ``[-1, input_dim]``."""
        x = _rows(inputs, self.block_dict["input"].input_dim)
        specs = {p: (0, -_INF, _INF) for p in self.dist_params}
        out, _ = E.net_forward(self.plan(), x, self.training, None, specs)
        return out

    def _num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


def _wants_graph(dist) -> bool:
    """An autograd graph is built only for a training-mode network that belongs to a model arena, under grad mode."""
    return torch.is_grad_enabled() and dist.training and getattr(dist, "_arena_owner", None) is not None


def _rows(t: torch.Tensor, width: int) -> torch.Tensor:
    nv.ensure_device(t.device)
    return nv.f32(t, "input").reshape(-1, width)


def _expand_pair(targets: torch.Tensor, mu: torch.Tensor, sigma: torch.Tensor):
    """Broadcast targets against parameters the way the reference's tensor arithmetic would and return
    (x_rows, mu_rows, sigma_rows, out_shape, B) with x indexed as row % B."""
    shape = torch.broadcast_shapes(targets.shape, mu.shape, sigma.shape)
    D = shape[-1]
    lead = shape[:-1]
    mu_r = mu.expand(shape).reshape(-1, D).contiguous()
    sg_r = sigma.expand(shape).reshape(-1, D).contiguous()
    rows = mu_r.shape[0]
    # cheap case: targets cover the trailing leading dims only ([B,D] or [1,B,D] against [S,B,D])
    t = targets
    while t.ndim > 2 and t.shape[0] == 1:
        t = t[0]
    if t.ndim == 2 and len(lead) >= 1 and t.shape[0] == lead[-1] and t.shape[1] == D:
        return t.contiguous(), mu_r, sg_r, lead, t.shape[0]
    x_r = targets.expand(shape).reshape(-1, D).contiguous()
    return x_r, mu_r, sg_r, lead, rows


class GaussianNN(torch.nn.Module):
    """Diagonal Gaussian whose mean and (softplus) standard deviation come from an MLP
    (reference: src/submodels.py:74-124)."""

    def __init__(self, input_dim, output_dim, sigma_fixed=1.0, sigma_lim=0.1, marginal_std_lim=None, average_max_std=1.0,
                 mu_upper_lim=5.0, mu_lower_lim=-3.0, learn_sigma=True, num_hidden_layers=2, num_neurons=50, dropout=True,
                 dropout_rate=0.5, batch_normalization=True, **_):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.learn_sigma, self.sigma_fixed = learn_sigma, sigma_fixed
        if marginal_std_lim is not None:
            self.sigma_lower_lim = marginal_std_lim
            print("USING MARGINAL_STD_LIM!")
        else:
            self.sigma_lower_lim = sigma_lim
            print("USING SIGMA_LIM!")
        self.sigma_upper_lim = average_max_std
        self.mu_upper_lim, self.mu_lower_lim = mu_upper_lim, mu_lower_lim
        self.parameterizer = ParameterizerNN(input_dim, output_dim, dist_params=["mu", "sigma"] if learn_sigma else ["mu"],
                                             num_hidden_layers=num_hidden_layers, num_neurons=num_neurons, dropout=dropout,
                                             dropout_rate=dropout_rate, batch_normalization=batch_normalization)
        self._plan = None

    def plan(self) -> E.NetPlan:
        if self._plan is None:
            object.__setattr__(self, "_plan", E.NetPlan(self))
        return self._plan

    def head_specs(self):
        specs = {"mu": (0, self.mu_lower_lim, self.mu_upper_lim)}
        if self.learn_sigma:
            specs["sigma"] = (1, self.sigma_lower_lim, self.sigma_upper_lim)
        return specs

    def _num_parameters(self):
        return self.parameterizer._num_parameters()

    def forward(self, inputs):
        x = _rows(inputs, self.input_dim)
        if _wants_graph(self):
            out = apply_net(self, x)
        else:
            out, _ = E.net_forward(self.plan(), x, self.training, None, self.head_specs())
        if not self.learn_sigma:
            out["sigma"] = torch.full_like(out["mu"], float(self.sigma_fixed))
        return out

    def rsample(self, param_dict=None, num_samples=1, **_):
        mu, sigma = param_dict["mu"], param_dict["sigma"]
        shape = torch.broadcast_shapes(mu.shape, sigma.shape)
        eps = torch.randn((num_samples, *shape), device=mu.device)
        if torch.is_grad_enabled() and (mu.requires_grad or sigma.requires_grad):
            return mu + sigma * eps          # keeps the graph; same arithmetic as the kernel
        L = shape[-1]
        mu_r, sg_r = mu.expand(shape).reshape(-1, L).contiguous(), sigma.expand(shape).reshape(-1, L).contiguous()
        nv.ensure_device(mu_r.device)
        z, _, _ = E.reparam_kl(mu_r, sg_r, eps.reshape(-1, L), None, num_samples, False, mu_r.shape[0])
        return z.view(num_samples, *shape)

    def sample(self, param_dict=None, num_samples=1, **_):
        return self.rsample(param_dict=param_dict, num_samples=num_samples)

    def log_likelihood(self, targets, param_dict=None):
        if torch.is_grad_enabled() and (param_dict["mu"].requires_grad or param_dict["sigma"].requires_grad):
            return log_prob("Normal", param_dict, targets).sum(dim=-1)
        x_r, mu_r, sg_r, lead, _ = _expand_pair(targets, param_dict["mu"], param_dict["sigma"])
        nv.ensure_device(mu_r.device)
        logp, _, _ = E.diag_loglik(nv.f32(x_r), nv.f32(mu_r), nv.f32(sg_r))
        return logp.view(lead)

    def kl_divergence(self, param_dict=None, prior_params={"mu": 0.0, "sigma": 1.0}):
        return kl_divergence("Normal", param_dict, prior_params)

    def create_covariance_matrix(self, param_dict):
        return torch.diag_embed(param_dict["sigma"] ** 2)

    def get_marginal_sigmas(self, param_dict):
        return param_dict["sigma"]


class DictionaryGaussian(torch.nn.Module):
    """Multivariate normal with covariance (Û diag σ)(Û diag σ)ᵀ + m²I, Û a learned column-normalised
    dictionary (reference: src/submodels.py:126-182)."""

    def __init__(self, input_dim, output_dim, vocab_size=100, sigma_lim=0.1, marginal_std_lim=0.01, average_max_std=1.0,
                 mu_upper_lim=5.0, mu_lower_lim=-3.0, num_hidden_layers=2, num_neurons=50, dropout=True, dropout_rate=0.5,
                 batch_normalization=True, **_):
        super().__init__()
        if vocab_size <= output_dim:
            raise ValueError("Vocabulary size should be larger than the output dimension. Please increase the vocabulary size.")
        if output_dim > 128:
            raise ValueError("The HIP dictionary-Gaussian kernels support output_dim <= 128 (registers up to 64, LDS beyond).")
        self.vocab_size, self.marginal_std_lim, self.output_dim = vocab_size, marginal_std_lim, output_dim
        self.input_dim = input_dim
        self.mu_upper_lim, self.mu_lower_lim = mu_upper_lim, mu_lower_lim
        self.sigma_lower_lim = sigma_lim
        self.sigma_upper_lim = (output_dim / vocab_size) ** .5 * (average_max_std - marginal_std_lim)
        if self.sigma_upper_lim < self.sigma_lower_lim:
            raise ValueError("Sigma upper limit is smaller than sigma lower limit. Please decresase the vocabulary size or the sigma lower limit.")
        self.parameterizer = ParameterizerNN(input_dim, {"mu": output_dim, "sigma": vocab_size}, dist_params=["mu", "sigma"],
                                             num_hidden_layers=num_hidden_layers, num_neurons=num_neurons, dropout=dropout,
                                             dropout_rate=dropout_rate, batch_normalization=batch_normalization)
        self.SigmaMapper = torch.nn.Parameter(torch.randn(output_dim, vocab_size), requires_grad=True)
        self._plan = None

    def plan(self) -> E.NetPlan:
        if self._plan is None:
            object.__setattr__(self, "_plan", E.NetPlan(self))
        return self._plan

    def head_specs(self):
        return {"mu": (0, self.mu_lower_lim, self.mu_upper_lim), "sigma": (2, self.sigma_lower_lim, self.sigma_upper_lim)}

    def _num_parameters(self):
        return self.parameterizer._num_parameters()

    def tables(self) -> E.DictTables:
        nv.ensure_device(self.SigmaMapper.device)
        return E.DictTables(self.SigmaMapper.detach().contiguous(), self.marginal_std_lim)

    def get_SigmaMapper(self):
        return self.tables().Uhat

    def forward(self, inputs):
        x = _rows(inputs, self.input_dim)
        if _wants_graph(self):
            return apply_net(self, x)
        out, _ = E.net_forward(self.plan(), x, self.training, None, self.head_specs())
        return out

    def rsample(self, param_dict=None, num_samples=1, **_):
        mu, sigma = param_dict["mu"].detach(), param_dict["sigma"].detach()
        eps = torch.randn((num_samples, *sigma.shape), device=sigma.device)
        scaled = (eps * sigma).reshape(-1, self.vocab_size).contiguous()
        rows = scaled.shape[0]
        mu_r = mu.expand(num_samples, *mu.shape).reshape(rows, self.output_dim).contiguous()
        Uh = self.get_SigmaMapper()
        out = torch.empty(rows, self.output_dim, device=mu.device)
        nv.call("btvae_matmul_nt_add", nv.ptr(scaled), self.vocab_size, nv.ptr(Uh), self.vocab_size, nv.ptr(mu_r), self.output_dim,
                nv.ptr(out), self.output_dim, rows, self.output_dim, self.vocab_size)
        return out.view(num_samples, *mu.shape)

    def sample(self, param_dict=None, num_samples=1, **_):
        return self.rsample(param_dict=param_dict, num_samples=num_samples)

    def log_likelihood(self, targets, param_dict=None):
        mu, sigma = param_dict["mu"], param_dict["sigma"]
        lead = torch.broadcast_shapes(targets.shape[:-1], mu.shape[:-1])
        D, V = self.output_dim, self.vocab_size
        mu_r = mu.expand(*lead, D).reshape(-1, D).contiguous()
        sg_r = sigma.expand(*lead, V).reshape(-1, V).contiguous()
        x_r, _, _, _, B = _expand_pair(targets, mu.expand(*lead, D), mu.expand(*lead, D))
        if torch.is_grad_enabled() and (mu.requires_grad or sigma.requires_grad):
            return dict_log_likelihood(self, nv.f32(x_r.detach()), mu_r, sg_r).view(lead)
        logp = E.dict_loglik_fwd(self.tables(), nv.f32(x_r), nv.f32(mu_r), (sg_r * sg_r).contiguous())
        return logp.view(lead)

    def kl_divergence(self, param_dict=None, prior_params={"mu": 0.0, "sigma": 1.0}):
        pass

    def create_covariance_matrix(self, param_dict):
        sigma = param_dict["sigma"]
        tab = self.tables()
        sg = sigma.reshape(-1, self.vocab_size).contiguous()
        Sp = E.dict_covariance_packed(tab, (sg * sg).contiguous())
        D = self.output_dim
        tri = torch.tril_indices(D, D, device=Sp.device)
        full = torch.zeros(Sp.shape[0], D, D, device=Sp.device)
        full[:, tri[0], tri[1]] = Sp
        full[:, tri[1], tri[0]] = Sp
        return full.view(*sigma.shape[:-1], D, D)

    def get_marginal_sigmas(self, param_dict):
        return torch.sqrt(torch.diagonal(self.create_covariance_matrix(param_dict), dim1=-2, dim2=-1))


def get_distribution_model(dist_type, **kwargs):
    """Factory with the reference's aliases (src/submodels.py:184-187)."""
    key = dist_type.lower()
    if key in ("gaussian", "gauss", "normal", "n", "g"):
        return GaussianNN(**kwargs)
    if key in ("dictionary-gaussian", "dict-gauss"):
        return DictionaryGaussian(**kwargs)
    raise NotImplementedError("Unknown distribution type: {}".format(dist_type))


def get_prior_params(dist_type, num_dims=1):
    """Standard-normal prior parameter tensors, scalar or list form (src/submodels.py:189-198)."""
    many = type(dist_type) is list
    kinds, dims = (dist_type, num_dims) if many else ([dist_type], [num_dims])
    out = []
    for k, d in zip(kinds, dims):
        if k.lower() not in ("gaussian", "gauss", "normal", "n", "g"):
            raise NotImplementedError("Unknown distribution type: {}".format(k))
        out.append({"mu": torch.zeros(d), "sigma": torch.ones(d)})
    return out if many else out[0]
