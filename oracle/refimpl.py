"""CPU oracle: a functional restatement of the reference's conditional-VAE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.
It is the checker the HIP kernels are compared against, never the thing measured or shipped.

Pinning: the reference ships no tests or golden vectors for this path, so the oracle is
pinned by vectors produced by the reference itself, imported in the authoring container by
``oracle/make_golden.py`` and committed under ``tests/golden/`` (``tests/test_oracle_golden.py``
checks every one of them against this file).

Everything here works on a flat ``dict[str, Tensor]`` that uses the reference's state-dict key
names, so a reference checkpoint can be fed to it directly.  All functions are dtype-agnostic:
pass float64 tensors to get an arbitration-grade result, float32 to mimic the reference.

Reference citations are ``path:line`` relative to the reference repository root.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
BN_EPS = 1e-5        # torch.nn.BatchNorm1d default, src/submodels.py:20
BN_MOMENTUM = 0.1    # torch.nn.BatchNorm1d default
GAUSS_ALIASES = ("gaussian", "gauss", "normal", "n", "g")            # src/submodels.py:185
DICT_ALIASES = ("dictionary-gaussian", "dict-gauss")                  # src/submodels.py:186


# ---------------------------------------------------------------------------------------------
# effective hyper-parameters (what the reference's **_ swallowing leaves in force)
# ---------------------------------------------------------------------------------------------
@dataclass
class DistSpec:
    """Effective settings of one distribution network after the reference's keyword swallowing
    (src/submodels.py:75,127: unknown keys such as ``total_max_std`` and ``resnet`` vanish)."""
    kind: str                 # "normal" | "dict"
    in_dim: int
    out_dim: int
    hidden: int = 50
    n_mid: int = 2
    bn: bool = True
    dropout: bool = True
    dropout_rate: float = 0.5
    learn_sigma: bool = True
    sigma_fixed: float = 1.0
    mu_lo: float = -3.0
    mu_hi: float = 5.0
    sig_lo: float = 0.1
    sig_hi: float = 1.0
    vocab: int = 0
    marginal_std_lim: float = 0.0

    @property
    def heads(self) -> List[Tuple[str, int]]:
        if self.kind == "dict":
            return [("mu", self.out_dim), ("sigma", self.vocab)]
        if self.learn_sigma:
            return [("mu", self.out_dim), ("sigma", self.out_dim)]
        return [("mu", self.out_dim)]


def dist_spec(in_dim: int, out_dim: int, cfg: dict) -> DistSpec:
    """Resolve a ``distribution_dict`` entry into effective numbers.

    GaussianNN defaults: src/submodels.py:75-90 (``marginal_std_lim`` overrides ``sigma_lim`` as the
    lower limit when given; the upper limit is ``average_max_std``).
    DictionaryGaussian defaults and limits: src/submodels.py:127-139."""
    kind = cfg["dist_type"].lower()
    if kind in GAUSS_ALIASES:
        msl = cfg.get("marginal_std_lim", None)
        return DistSpec(
            kind="normal", in_dim=in_dim, out_dim=out_dim,
            hidden=cfg.get("num_neurons", 50), n_mid=cfg.get("num_hidden_layers", 2),
            bn=cfg.get("batch_normalization", True), dropout=cfg.get("dropout", True),
            dropout_rate=cfg.get("dropout_rate", 0.5), learn_sigma=cfg.get("learn_sigma", True),
            sigma_fixed=cfg.get("sigma_fixed", 1.0),
            mu_lo=cfg.get("mu_lower_lim", -3.0), mu_hi=cfg.get("mu_upper_lim", 5.0),
            sig_lo=(msl if msl is not None else cfg.get("sigma_lim", 0.1)),
            sig_hi=cfg.get("average_max_std", 1.0))
    if kind in DICT_ALIASES:
        vocab = cfg.get("vocab_size", 100)
        if vocab <= out_dim:
            raise ValueError("Vocabulary size should be larger than the output dimension.")
        msl = cfg.get("marginal_std_lim", 0.01)
        hi = (out_dim / vocab) ** 0.5 * (cfg.get("average_max_std", 1.0) - msl)
        lo = cfg.get("sigma_lim", 0.1)
        if hi < lo:
            raise ValueError("Sigma upper limit is smaller than sigma lower limit.")
        return DistSpec(
            kind="dict", in_dim=in_dim, out_dim=out_dim,
            hidden=cfg.get("num_neurons", 50), n_mid=cfg.get("num_hidden_layers", 2),
            bn=cfg.get("batch_normalization", True), dropout=cfg.get("dropout", True),
            dropout_rate=cfg.get("dropout_rate", 0.5),
            mu_lo=cfg.get("mu_lower_lim", -3.0), mu_hi=cfg.get("mu_upper_lim", 5.0),
            sig_lo=lo, sig_hi=hi, vocab=vocab, marginal_std_lim=msl)
    raise NotImplementedError("Unknown distribution type: {}".format(cfg["dist_type"]))


@dataclass
class ModelSpec:
    input_dim: int
    cond_dim: int
    latent_dim: int
    enc: DistSpec
    dec: DistSpec


def model_spec(input_dim: int, cond_dim: int, model_cfg: dict) -> ModelSpec:
    """CVAE wiring, src/vae_models.py:253-258: encoder eats x‖c, decoder eats z‖c."""
    L = model_cfg.get("latent_dim", 10)
    dd = model_cfg["distribution_dict"]
    return ModelSpec(input_dim, cond_dim, L,
                     dist_spec(input_dim + cond_dim, L, dd["posterior"]),
                     dist_spec(L + cond_dim, input_dim, dd["likelihood"]))


# ---------------------------------------------------------------------------------------------
# network walk: the ordered list of layers of one ParameterizerNN
# ---------------------------------------------------------------------------------------------
@dataclass
class LayerRef:
    """One Linear(+ReLU+BN) unit, addressed by its state-dict prefix."""
    prefix: str          # e.g. "encoder.parameterizer.block_dict.input.middle_layers.0"
    in_dim: int
    out_dim: int
    act_bn: bool         # Linear -> ReLU -> [BN] -> [Dropout]; False = bare Linear (head output)
    residual: bool       # h = layer(h) + h  (src/submodels.py:32)


def network_layers(net: str, spec: DistSpec) -> Tuple[List[LayerRef], Dict[str, List[LayerRef]]]:
    """Trunk and per-head layer lists of ``ParameterizerNN`` (src/submodels.py:41-69) built from
    ``NNBlock`` (src/submodels.py:6-35).  The trunk block's output Linear gets ReLU/BN/Dropout
    appended (src/submodels.py:49-51); heads are ``num_hidden_layers=0`` blocks."""
    root = f"{net}.parameterizer.block_dict"
    H = spec.hidden
    trunk = [LayerRef(f"{root}.input.input_layer", spec.in_dim, H, True, False)]
    for i in range(spec.n_mid):
        trunk.append(LayerRef(f"{root}.input.middle_layers.{i}", H, H, True, True))
    trunk.append(LayerRef(f"{root}.input.output_layer", H, H, True, False))
    heads = {}
    for name, width in spec.heads:
        heads[name] = [LayerRef(f"{root}.{name}.input_layer", H, H, True, False),
                       LayerRef(f"{root}.{name}.output_layer", H, width, False, False)]
    return trunk, heads


def parameter_keys(spec: ModelSpec) -> List[str]:
    """Trainable tensors in ``self.parameters()`` order of the reference module tree: encoder
    parameterizer, then decoder ``SigmaMapper`` (registered before the parameterizer's module
    parameters are visited, since direct parameters come first), then decoder parameterizer."""
    keys: List[str] = []
    for net, ds in (("encoder", spec.enc), ("decoder", spec.dec)):
        if ds.kind == "dict":
            keys.append(f"{net}.SigmaMapper")
        trunk, heads = network_layers(net, ds)
        for lay in trunk + [l for h in heads.values() for l in h]:
            keys += [f"{lay.prefix}.0.weight", f"{lay.prefix}.0.bias"]
            if lay.act_bn and ds.bn:
                keys += [f"{lay.prefix}.2.weight", f"{lay.prefix}.2.bias"]
    return keys


def init_state(spec: ModelSpec, seed: int, dtype=torch.float32) -> Dict[str, Tensor]:
    """Fresh state dict with torch's default Linear/BatchNorm initialisation and a standard-normal
    ``SigmaMapper`` (src/submodels.py:146).  Not bit-identical to a reference construction (the
    reference consumes RNG for throw-away networks first, src/vae_models.py:29-30,255-256); tests that
    need identical weights load a state dict instead."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for net, ds in (("encoder", spec.enc), ("decoder", spec.dec)):
        trunk, heads = network_layers(net, ds)
        for lay in trunk + [l for h in heads.values() for l in h]:
            bound = 1.0 / math.sqrt(lay.in_dim)
            sd[f"{lay.prefix}.0.weight"] = (torch.rand(lay.out_dim, lay.in_dim, generator=g, dtype=torch.float64) * 2 - 1).mul(bound).to(dtype)
            sd[f"{lay.prefix}.0.bias"] = (torch.rand(lay.out_dim, generator=g, dtype=torch.float64) * 2 - 1).mul(bound).to(dtype)
            if lay.act_bn and ds.bn:
                sd[f"{lay.prefix}.2.weight"] = torch.ones(lay.out_dim, dtype=dtype)
                sd[f"{lay.prefix}.2.bias"] = torch.zeros(lay.out_dim, dtype=dtype)
                sd[f"{lay.prefix}.2.running_mean"] = torch.zeros(lay.out_dim, dtype=dtype)
                sd[f"{lay.prefix}.2.running_var"] = torch.ones(lay.out_dim, dtype=dtype)
                sd[f"{lay.prefix}.2.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
        if ds.kind == "dict":
            sd[f"{net}.SigmaMapper"] = torch.randn(ds.out_dim, ds.vocab, generator=g, dtype=torch.float64).to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------
# layer math
# ---------------------------------------------------------------------------------------------
def linear_relu_bn(sd: Dict[str, Tensor], lay: LayerRef, h: Tensor, bn: bool, train: bool,
                   bn_updates: Optional[Dict[str, Tensor]] = None,
                   dropout_mask: Optional[Tensor] = None) -> Tensor:
    """``Sequential(Linear, ReLU, [BatchNorm1d], [Dropout])`` then the optional residual add
    (src/submodels.py:15-25,28-33).  In train mode BatchNorm normalises with the biased batch
    variance and moves the running statistics with momentum 0.1 using the unbiased variance;
    new running values are returned through ``bn_updates`` (the state dict itself is not mutated).
    ``dropout_mask`` (already scaled by 1/(1-p)) is applied after BatchNorm when given."""
    W, b = sd[f"{lay.prefix}.0.weight"], sd[f"{lay.prefix}.0.bias"]
    y = h @ W.t() + b
    if not lay.act_bn:
        return y
    y = torch.relu(y)
    if bn:
        gam, bet = sd[f"{lay.prefix}.2.weight"], sd[f"{lay.prefix}.2.bias"]
        if train:
            n = y.shape[0]
            mean = y.mean(0)
            var_b = ((y - mean) ** 2).mean(0)
            if bn_updates is not None:
                var_u = var_b * (n / max(n - 1, 1))
                rm, rv = sd[f"{lay.prefix}.2.running_mean"], sd[f"{lay.prefix}.2.running_var"]
                bn_updates[f"{lay.prefix}.2.running_mean"] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean).detach()
                bn_updates[f"{lay.prefix}.2.running_var"] = ((1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_u).detach()
                bn_updates[f"{lay.prefix}.2.num_batches_tracked"] = sd[f"{lay.prefix}.2.num_batches_tracked"] + 1
        else:
            mean, var_b = sd[f"{lay.prefix}.2.running_mean"], sd[f"{lay.prefix}.2.running_var"]
        y = (y - mean) / torch.sqrt(var_b + BN_EPS) * gam + bet
    if dropout_mask is not None:
        y = y * dropout_mask
    return y + h if lay.residual else y


def parameterizer_forward(sd: Dict[str, Tensor], net: str, ds: DistSpec, inputs: Tensor, train: bool,
                          bn_updates: Optional[Dict[str, Tensor]] = None,
                          keep_flags: Optional[List[Tensor]] = None) -> Dict[str, Tensor]:
    """``ParameterizerNN.forward`` (src/submodels.py:63-69): flatten to rows, shared trunk, then
    one two-layer head per distribution parameter.  Train-mode dropout (src/submodels.py:23-25,51)
    takes its Bernoulli keep flags from the front of ``keep_flags`` (one 0/1 tensor per
    Linear-ReLU-BN unit, in execution order); with ``keep_flags=None`` dropout is inactive."""
    trunk, heads = network_layers(net, ds)
    h = inputs.reshape(-1, ds.in_dim)

    def mask_for(lay):
        if not (train and ds.dropout and lay.act_bn and keep_flags is not None):
            return None
        return keep_flags.pop(0).to(h.dtype) / (1.0 - ds.dropout_rate)

    for lay in trunk:
        h = linear_relu_bn(sd, lay, h, ds.bn, train, bn_updates, mask_for(lay))
    out = {}
    for name, lays in heads.items():
        t = h
        for lay in lays:
            t = linear_relu_bn(sd, lay, t, ds.bn, train, bn_updates, mask_for(lay))
        out[name] = t
    return out


def softplus_thr(x: Tensor, threshold: float = 5.0) -> Tensor:
    """``softplus(x, beta=1, threshold=5)`` (src/utils.py:10): identity above the threshold."""
    return torch.where(x > threshold, x, torch.log1p(torch.exp(torch.minimum(x, torch.full_like(x, threshold)))))


def dist_forward(sd: Dict[str, Tensor], net: str, ds: DistSpec, inputs: Tensor, train: bool,
                 bn_updates: Optional[Dict[str, Tensor]] = None,
                 keep_flags: Optional[List[Tensor]] = None) -> Dict[str, Tensor]:
    """``GaussianNN.forward`` (src/submodels.py:100-106) / ``DictionaryGaussian.forward``
    (src/submodels.py:154-158): clamp mu; sigma = clamp(softplus(.)) for the Gaussian,
    clamp(relu(.)) for the dictionary Gaussian."""
    raw = parameterizer_forward(sd, net, ds, inputs, train, bn_updates, keep_flags)
    mu = raw["mu"].clamp(ds.mu_lo, ds.mu_hi)
    if ds.kind == "dict":
        sigma = torch.relu(raw["sigma"]).clamp(ds.sig_lo, ds.sig_hi)
    elif ds.learn_sigma:
        sigma = softplus_thr(raw["sigma"]).clamp(ds.sig_lo, ds.sig_hi)
    else:
        sigma = torch.ones_like(mu) * ds.sigma_fixed
    return {"mu": mu, "sigma": sigma}


def normal_log_prob(t: Tensor, mu: Tensor, sigma: Tensor) -> Tensor:
    """Element-wise Normal log density (src/utils.py:21-23)."""
    return -0.5 * math.log(2 * math.pi) - torch.log(sigma) - 0.5 * ((t - mu) / sigma) ** 2


def normal_kl(mu: Tensor, sigma: Tensor, pmu, psigma) -> Tensor:
    """Element-wise KL(N(mu,sigma) || N(pmu,psigma)) (src/utils.py:29-33)."""
    ratio = (sigma / psigma) ** 2
    return 0.5 * (ratio + ((mu - pmu) / psigma) ** 2 - 1 - torch.log(ratio))


def sigma_mapper(U: Tensor) -> Tensor:
    """Column-normalised dictionary (src/utils.py:18-19, src/submodels.py:151-152)."""
    return U / torch.linalg.norm(U, dim=0, keepdim=True)


def dict_covariance(U: Tensor, sigma: Tensor, marginal_std_lim: float) -> Tensor:
    """Sigma = (Û diag σ)(Û diag σ)ᵀ + m² I (src/submodels.py:176-179)."""
    Uh = sigma_mapper(U)
    S = Uh * sigma[..., None, :]
    D = U.shape[0]
    return S @ S.transpose(-1, -2) + torch.eye(D, dtype=U.dtype) * marginal_std_lim ** 2


def mvn_log_prob(x: Tensor, mu: Tensor, Sigma: Tensor) -> Tensor:
    """log N(x; mu, Sigma) by Cholesky, as ``MultivariateNormal(covariance_matrix=).log_prob``
    computes it (src/submodels.py:169-171): −½(D log 2π + ‖L⁻¹(x−μ)‖²) − Σ log diag L."""
    L = torch.linalg.cholesky(Sigma)
    r = (x - mu).unsqueeze(-1)
    y = torch.linalg.solve_triangular(L, r, upper=False).squeeze(-1)
    D = mu.shape[-1]
    return -0.5 * (D * math.log(2 * math.pi) + (y ** 2).sum(-1)) - torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)


def decoder_log_likelihood(sd: Dict[str, Tensor], ds: DistSpec, x: Tensor, params: Dict[str, Tensor]) -> Tensor:
    """``decoder.log_likelihood(targets, params)``: dictionary-Gaussian MVN (src/submodels.py:169-171)
    or summed diagonal Normal (src/submodels.py:114-115)."""
    if ds.kind == "dict":
        Sigma = dict_covariance(sd["decoder.SigmaMapper"], params["sigma"], ds.marginal_std_lim)
        return mvn_log_prob(x, params["mu"], Sigma)
    return normal_log_prob(x, params["mu"], params["sigma"]).sum(-1)


def dict_rsample(U: Tensor, mu: Tensor, sigma: Tensor, eps: Tensor) -> Tensor:
    """``DictionaryGaussian.rsample`` (src/submodels.py:160-164): mu + Û (eps ⊙ sigma); no m² term."""
    return mu + (sigma_mapper(U) @ (eps * sigma)[..., None]).squeeze(-1)


# ---------------------------------------------------------------------------------------------
# model forward / loss
# ---------------------------------------------------------------------------------------------
def cvae_forward(sd: Dict[str, Tensor], spec: ModelSpec, x: Tensor, c: Tensor, eps: Tensor, train: bool,
                 bn_updates: Optional[Dict[str, Tensor]] = None, keep_flags: Optional[List[Tensor]] = None):
    """``CVAE.forward`` (src/vae_models.py:260-280).

    2-D conditions ``c[B,C]``: encoder on B rows, ``z = mu + sigma*eps`` with ``eps[S,B,L]``, decoder
    on S·B rows of ``z ‖ c``.  3-D conditions ``c[B,S,C]``: encoder on S·B rows of ``x ‖ c[s]``,
    posterior parameters ``[S,B,L]``, one draw per row (``eps[S,B,L]``).  Returns
    (likelihood params ``[S,B,·]``, posterior params, z)."""
    if c.ndim == 2:
        S, B = eps.shape[0], x.shape[0]
        post = dist_forward(sd, "encoder", spec.enc, torch.cat((x, c), -1), train, bn_updates, keep_flags)
        z = post["mu"] + post["sigma"] * eps
        dec_in = torch.cat((z, c.unsqueeze(0).expand(S, B, c.shape[-1])), 2)
    elif c.ndim == 3:
        S, B = eps.shape[0], x.shape[0]
        if c.shape[1] != S:
            raise ValueError("Condition samples must be equal to number of posterior MC samples.")
        ct = c.transpose(1, 0)
        post = dist_forward(sd, "encoder", spec.enc, torch.cat((x.unsqueeze(0).expand(S, B, x.shape[-1]), ct), -1), train, bn_updates, keep_flags)
        post = {k: v.view(S, B, -1) for k, v in post.items()}
        z = post["mu"] + post["sigma"] * eps
        dec_in = torch.cat((z, ct), -1)
    else:
        raise ValueError("Invalid shape for conditions. Must be 2D or 3D tensor.")
    lik = dist_forward(sd, "decoder", spec.dec, dec_in, train, bn_updates, keep_flags)
    lik = {k: v.view(S, B, -1) for k, v in lik.items()}
    return lik, post, z


def elbo_terms(sd: Dict[str, Tensor], spec: ModelSpec, x: Tensor, lik: Dict[str, Tensor], post: Dict[str, Tensor],
               beta: float = 1.0) -> Dict[str, Tensor]:
    """``VAE.loss`` (src/vae_models.py:70-74): rll = mean over all S·B log-likelihoods, kl = mean of
    per-row KL sums (prior N(0,1), src/submodels.py:195), loss = −(rll − β·kl), elbo = rll − kl."""
    rll = decoder_log_likelihood(sd, spec.dec, x, lik).mean()
    kl = normal_kl(post["mu"], post["sigma"], 0.0, 1.0).sum(-1).mean()
    return {"loss": -(rll - beta * kl), "elbo": rll - kl, "rll": rll, "kl": kl}


# ---------------------------------------------------------------------------------------------
# optimiser step (clip + Adam with L2), restated from torch's documented algorithm
# ---------------------------------------------------------------------------------------------
@dataclass
class AdamState:
    step: int = 0
    m: Dict[str, Tensor] = field(default_factory=dict)
    v: Dict[str, Tensor] = field(default_factory=dict)


def clip_and_adam(sd: Dict[str, Tensor], grads: Dict[str, Tensor], st: AdamState, lr: float, weight_decay: float,
                  max_norm: Optional[float], betas=(0.9, 0.999), eps: float = 1e-8) -> float:
    """``clip_grad_norm_(parameters, max_norm)`` then ``Adam.step`` with L2 weight decay
    (src/vae_models.py:323-324, :103).  Returns the pre-clip global gradient norm."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    coef = 1.0
    if max_norm is not None:
        coef = min(max_norm / (total + 1e-6), 1.0)
    st.step += 1
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** st.step, 1 - b2 ** st.step
    for k, g in grads.items():
        p = sd[k]
        g = g * coef + weight_decay * p
        if k not in st.m:
            st.m[k], st.v[k] = torch.zeros_like(p), torch.zeros_like(p)
        st.m[k] = b1 * st.m[k] + (1 - b1) * g
        st.v[k] = b2 * st.v[k] + (1 - b2) * g * g
        denom = st.v[k].sqrt() / math.sqrt(bc2) + eps
        sd[k] = (p - (lr / bc1) * st.m[k] / denom).detach()
    return total


def clip_and_sgd(sd: Dict[str, Tensor], grads: Dict[str, Tensor], lr: float, max_norm: Optional[float]) -> float:
    """``clip_grad_norm_(parameters, max_norm)`` then a plain ``torch.optim.SGD(lr)`` step (no momentum, no decay):
    ``train_core`` steps whatever optimiser it is handed (src/vae_models.py:317-325).  Returns the pre-clip norm."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    coef = min(max_norm / (total + 1e-6), 1.0) if max_norm is not None else 1.0
    for k, g in grads.items():
        sd[k] = (sd[k] - lr * (g * coef)).detach()
    return total


def train_step(sd: Dict[str, Tensor], spec: ModelSpec, st: Optional[AdamState], x: Tensor, c: Tensor, eps: Tensor,
               lr: float, weight_decay: float, beta: float = 1.0, max_norm: Optional[float] = 1.0,
               keep_flags: Optional[List[Tensor]] = None, optimizer: str = "adam"):
    """One ``CVAE.train_core`` (src/vae_models.py:317-325): forward in train mode, loss, backward
    (autograd on the restated forward), global-norm clip, Adam (or plain SGD with ``optimizer="sgd"``).
    Mutates ``sd`` and ``st``.  Returns (loss dict of floats, gradient dict, pre-clip norm)."""
    keys = parameter_keys(spec)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    upd: Dict[str, Tensor] = {}
    lik, post, _ = cvae_forward(work, spec, x, c, eps, True, upd, keep_flags)
    terms = elbo_terms(work, spec, x, lik, post, beta)
    gl = torch.autograd.grad(terms["loss"], [leaves[k] for k in keys])
    grads = {k: g.detach() for k, g in zip(keys, gl)}
    sd.update(upd)
    if optimizer == "sgd":
        norm = clip_and_sgd(sd, grads, lr, max_norm)
    else:
        norm = clip_and_adam(sd, grads, st, lr, weight_decay, max_norm)
    return {k: float(v.detach()) for k, v in terms.items()}, grads, norm


def validate(sd: Dict[str, Tensor], spec: ModelSpec, batches: Sequence[Tuple[Tensor, Tensor]], eps_list: Sequence[Tensor],
             n_dataset: int, beta: float = 1.0) -> Dict[str, float]:
    """``CVAE.validate`` (src/vae_models.py:327-337): eval-mode forward per batch, loss terms weighted by the batch's
    row count and divided by the dataset length.  ``eps_list[i]`` holds batch i's latent noise (any shape with
    S*B*L elements)."""
    tot = {"elbo": 0.0, "rll": 0.0, "kl": 0.0}
    for (x, c), eps in zip(batches, eps_list):
        S = eps.numel() // (x.shape[0] * spec.latent_dim)
        lik, post, _ = cvae_forward(sd, spec, x, c, eps.reshape(S, x.shape[0], spec.latent_dim), False)
        terms = elbo_terms(sd, spec, x, lik, post, beta)
        for k in tot:
            tot[k] += float(terms[k]) * x.shape[0]
    return {k: v / n_dataset for k, v in tot.items()}


class PlateauMax:
    """``ReduceLROnPlateau(mode="max", threshold_mode="abs")`` as ``fit`` builds it (src/vae_models.py:181), restated
    from torch's documented rule: a value is better when it exceeds best + threshold; after more than ``patience``
    non-improving values in a row the rate is multiplied by ``factor`` (not below ``min_lr``, and only if the change
    exceeds 1e-8) and the count restarts."""

    def __init__(self, lr: float, threshold: float = 1e-4, factor: float = 0.1, patience: int = 10, min_lr: float = 0.0):
        self.lr, self.threshold, self.factor, self.patience, self.min_lr = lr, threshold, factor, patience, min_lr
        self.best, self.bad = -math.inf, 0

    def step(self, value: float) -> float:
        if value > self.best + self.threshold:
            self.best, self.bad = value, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = max(self.lr * self.factor, self.min_lr)
            if self.lr - new > 1e-8:
                self.lr = new
            self.bad = 0
        return self.lr


class EarlyStop:
    """``EarlyStopping`` (src/utils.py:156-180): the first score only sets the best; later a score must beat
    best + delta, else a counter runs up to ``patience``."""

    def __init__(self, patience: int = 5, delta: float = 0.1):
        self.patience, self.delta, self.best, self.count, self.stop = patience, delta, None, 0, False

    def __call__(self, score: float):
        if self.best is None:
            self.best = score
        elif self.stop:
            return
        elif score > self.best + self.delta:
            self.best, self.count = score, 0
        else:
            self.count += 1
            if self.count >= self.patience:
                self.stop = True


def fit_replay(sd: Dict[str, Tensor], spec: ModelSpec, train: Sequence[Tuple[Tensor, Tensor]], val: Sequence[Tuple[Tensor, Tensor]],
               n_val: int, draws: Sequence[Tensor], lr: float, weight_decay: float, beta: float, max_norm: Optional[float],
               epochs: int, validation_freq: int, plateau: Optional[dict], earlystopping: Optional[dict]):
    """The cadence of ``VAE.fit`` (src/vae_models.py:188-235) over fixed batches with the latent noise taken from
    ``draws`` in call order: ``train_core`` per batch; every ``validation_freq`` iterations ``validate`` (beta 1, :204),
    whose ELBO drives the plateau rule and early stopping; a raised stop ends training after the current epoch (:235).
    Returns (train loss terms per iteration, [elbo, rll, kl, iteration] per validation, learning rate after each
    validation)."""
    st = AdamState()
    sched = PlateauMax(lr, **plateau) if plateau is not None else None
    stopper = EarlyStop(**earlystopping) if earlystopping is not None else None
    it = iter(draws)
    L = spec.latent_dim
    train_log, val_log, lr_log = [], [], []
    for _ in range(epochs):
        for x, c in train:
            eps = next(it)
            terms, _, _ = train_step(sd, spec, st, x, c, eps.reshape(-1, x.shape[0], L), lr=(sched.lr if sched else lr),
                                     weight_decay=weight_decay, beta=beta, max_norm=max_norm)
            train_log.append([terms[k] for k in ("loss", "elbo", "rll", "kl")])
            if len(train_log) % validation_freq == 0 and val:
                v = validate(sd, spec, val, [next(it) for _ in val], n_val, beta=1.0)
                val_log.append([v["elbo"], v["rll"], v["kl"], len(train_log)])
                if sched is not None:
                    lr_log.append(sched.step(v["elbo"]))
                if stopper is not None:
                    stopper(v["elbo"])
        if stopper is not None and stopper.stop:
            break
    return train_log, val_log, lr_log


# ---------------------------------------------------------------------------------------------
# Monte-Carlo scoring
# ---------------------------------------------------------------------------------------------
def iw_loglikelihood(sd: Dict[str, Tensor], spec: ModelSpec, x: Tensor, c: Tensor, eps: Tensor):
    """K-sample importance-weighted log-likelihood with deterministic conditions
    (src/testing_lib.py:6-16 over :82-125; scripts/collect_testing_results.py:88-92):
    encoder once in eval mode, z_k = mu + sigma*eps_k, decoder log-likelihood per sample,
    ll = −log K + logsumexp_k(−log q(z_k|x,c) + log p(x|z_k,c) + log p(z_k)).
    Returns (ll[N], rlls[K,N], kl[N])."""
    K, N = eps.shape[0], x.shape[0]
    post = dist_forward(sd, "encoder", spec.enc, torch.cat((x, c), -1), False)
    z = post["mu"] + post["sigma"] * eps
    dec_in = torch.cat((z, c.unsqueeze(0).expand(K, N, c.shape[-1])), -1)
    lik = dist_forward(sd, "decoder", spec.dec, dec_in, False)
    lik = {k: v.view(K, N, -1) for k, v in lik.items()}
    rlls = decoder_log_likelihood(sd, spec.dec, x, lik)
    logq = normal_log_prob(z, post["mu"], post["sigma"]).sum(-1)
    logp = normal_log_prob(z, torch.zeros_like(z), torch.ones_like(z)).sum(-1)
    ll = -math.log(K) + torch.logsumexp(-logq + rlls + logp, dim=0)
    kl = normal_kl(post["mu"], post["sigma"], 0.0, 1.0).sum(-1)
    return ll, rlls, kl


def test_metrics(sd: Dict[str, Tensor], spec: ModelSpec, x: Tensor, c: Tensor, eps: Tensor, eps_x: Tensor) -> Dict[str, Tensor]:
    """Per-profile test metrics of scripts/collect_testing_results.py:84-113: importance-weighted log-likelihood, KL, and
    the l_p errors (p = 1, 2) of one posterior-predictive sample per latent draw,
    err_p[n] = (mean_k sum_d |x_s[k,n,d] − x[n,d]|^p)^(1/p) with x_s = mu + Û(eps_x ⊙ sigma) (src/submodels.py:160-164)."""
    K, N = eps.shape[0], x.shape[0]
    ll, _, kl = iw_loglikelihood(sd, spec, x, c, eps)
    post = dist_forward(sd, "encoder", spec.enc, torch.cat((x, c), -1), False)
    z = post["mu"] + post["sigma"] * eps
    lik = dist_forward(sd, "decoder", spec.dec, torch.cat((z, c.unsqueeze(0).expand(K, N, c.shape[-1])), -1), False)
    lik = {k: v.view(K, N, -1) for k, v in lik.items()}
    if spec.dec.kind == "dict":
        xs = dict_rsample(sd["decoder.SigmaMapper"], lik["mu"], lik["sigma"], eps_x)
    else:
        xs = lik["mu"] + lik["sigma"] * eps_x
    diff = (xs - x.unsqueeze(0)).abs()
    return {"loglikelihood": ll, "kl": kl, "l_1": diff.sum(-1).mean(0), "l_2": (diff ** 2).sum(-1).mean(0).sqrt()}


test_metrics.__test__ = False


def prior_sample_loglik(sd: Dict[str, Tensor], spec: ModelSpec, x: Tensor, c: Tensor, z: Tensor) -> Tensor:
    """Prior-sample scoring: decoder on z~N(0,I) ‖ c, log p(x|z,c) per sample
    (src/vae_models.py:283-290 followed by ``decoder.log_likelihood``). ``z[K,B,L]`` → ``[K,B]``."""
    K, B = z.shape[0], z.shape[1]
    dec_in = torch.cat((z, c.unsqueeze(0).expand(K, B, c.shape[-1])), -1)
    lik = dist_forward(sd, "decoder", spec.dec, dec_in, False)
    lik = {k: v.view(K, B, -1) for k, v in lik.items()}
    return decoder_log_likelihood(sd, spec.dec, x, lik)


# ---------------------------------------------------------------------------------------------
# conditional imputation
# ---------------------------------------------------------------------------------------------
def split_indices(D: int, observed: Sequence[int]) -> Tuple[List[int], List[int]]:
    """Observed index list as given, unobserved = ascending complement (src/imputation_lib.py:19)."""
    obs = [int(i) for i in observed]
    return obs, [i for i in range(D) if i not in obs]


def conditional_params(U: Tensor, ds: DistSpec, mu: Tensor, sigma: Tensor, x_obs: Tensor, observed: Sequence[int]):
    """Gaussian conditioning of the dictionary-Gaussian decoder output on observed coordinates
    (src/imputation_lib.py:21-35): mean_u + Σ_uo Σ_oo⁻¹ (x_o − μ_o), Σ_uu − Σ_uo Σ_oo⁻¹ Σ_uoᵀ."""
    obs, un = split_indices(ds.out_dim, observed)
    Sig = dict_covariance(U, sigma, ds.marginal_std_lim)
    Soo = Sig[..., obs, :][..., :, obs]
    Suo = Sig[..., un, :][..., :, obs]
    Suu = Sig[..., un, :][..., :, un]
    A = Suo @ torch.linalg.inv(Soo)
    mean = mu[..., un] + (A @ (x_obs - mu[..., obs])[..., None])[..., 0]
    cov = Suu - A @ Suo.transpose(-1, -2)
    return mean, cov


def pseudo_gibbs(sd: Dict[str, Tensor], spec: ModelSpec, x0: Tensor, missing_mask: Tensor, c: Tensor,
                 eps_z: Tensor, eps_x: Tensor, num_samples: int):
    """Pseudo-Gibbs chain with injected noise (src/imputation_lib.py:59-98, deterministic conditions).

    ``eps_z[T,B,L]`` drives the latent draw of iteration t, ``eps_x[T,B,U]`` the conditional draw.
    Integer/mask semantics kept: observed indices come from row 0 of the mask only (:63), ring slot
    is ``t % num_samples`` (:87), and the write-back ``x[mask] = samples.flatten()`` is a row-major
    masked scatter (:92).  Eval-mode networks (the callers load a model, which leaves it in eval mode).
    Returns (cond_samples[num_samples,B,U], cond_mean, cond_cov, x_final)."""
    x = x0.clone()
    T, B = eps_z.shape[0], x.shape[0]
    observed = torch.where(~missing_mask[0])[0].tolist()
    obs, un = split_indices(spec.input_dim, observed)
    Un = len(un)
    cs = torch.zeros(num_samples, B, Un, dtype=x.dtype)
    cm = torch.zeros(num_samples, B, Un, dtype=x.dtype)
    cc = torch.zeros(num_samples, B, Un, Un, dtype=x.dtype)
    U = sd["decoder.SigmaMapper"]
    for t in range(T):
        lik, _, _ = cvae_forward(sd, spec, x, c, eps_z[t:t + 1], False)
        mean, cov = conditional_params(U, spec.dec, lik["mu"][0], lik["sigma"][0], x[:, obs], obs)
        Lc = torch.linalg.cholesky(cov)
        samp = mean + (Lc @ eps_x[t][..., None])[..., 0]
        slot = t % num_samples
        cs[slot], cm[slot], cc[slot] = samp, mean, cov
        x[missing_mask] = samp.flatten()
    return cs, cm, cc, x


def metropolis_within_gibbs(sd: Dict[str, Tensor], spec: ModelSpec, x0: Tensor, missing_mask: Tensor, c: Tensor,
                            eps_z0: Tensor, eps_prop: Tensor, eps_x: Tensor, accept_u: Tensor, num_samples: int):
    """Metropolis-within-Gibbs sweeps with injected noise (src/imputation_lib.py:100-156, deterministic conditions,
    dictionary-Gaussian decoder, eval-mode networks).

    State z starts as a posterior draw (``eps_z0[B,L]``, :109).  Sweep t: posterior q(z|x,c) from the current x
    (:125), decoder at the current z (:127), proposal z' = mu + sigma*eps_prop[t] from the same posterior (:133),
    weights w(z) = log p(x|z) + log p(z) - log q(z|x) (:137-139), accept row b iff ``accept_u[t,b] < min(1, exp(w' - w))``
    (:143-147), then one conditional draw of the missing coordinates from the decoder at the (possibly updated) z
    (``eps_x[t]``, :151) written back by the row-major masked scatter (:153); ring slot ``t % num_samples``.
    Returns (cond_samples, cond_mean, cond_cov, accepted[T,B] bool, x_final)."""
    x = x0.clone()
    T, B = eps_prop.shape[0], x.shape[0]
    observed = torch.where(~missing_mask[0])[0].tolist()
    obs, un = split_indices(spec.input_dim, observed)
    Un = len(un)
    cs = torch.zeros(num_samples, B, Un, dtype=x.dtype)
    cm = torch.zeros(num_samples, B, Un, dtype=x.dtype)
    cc = torch.zeros(num_samples, B, Un, Un, dtype=x.dtype)
    U = sd["decoder.SigmaMapper"]
    accepted = torch.zeros(T, B, dtype=torch.bool)

    def posterior():
        return dist_forward(sd, "encoder", spec.enc, torch.cat((x, c), -1), False)

    def decode(zz):
        return dist_forward(sd, "decoder", spec.dec, torch.cat((zz, c), -1), False)

    def weight(zz, lik, post):
        return (decoder_log_likelihood(sd, spec.dec, x, lik)
                + normal_log_prob(zz, torch.zeros_like(zz), torch.ones_like(zz)).sum(-1)
                - normal_log_prob(zz, post["mu"], post["sigma"]).sum(-1))

    p0 = posterior()
    z = p0["mu"] + p0["sigma"] * eps_z0
    for t in range(T):
        post = posterior()
        lik = decode(z)
        z_prop = post["mu"] + post["sigma"] * eps_prop[t]
        lik_prop = decode(z_prop)
        log_ratio = weight(z_prop, lik_prop, post) - weight(z, lik, post)
        rho = torch.minimum(torch.ones(()), torch.exp(log_ratio))
        acc = accept_u[t] < rho
        accepted[t] = acc
        z = torch.where(acc[:, None], z_prop, z)
        lik = {k: torch.where(acc[:, None], lik_prop[k], lik[k]) for k in lik}
        mean, cov = conditional_params(U, spec.dec, lik["mu"], lik["sigma"], x[:, obs], obs)
        samp = mean + (torch.linalg.cholesky(cov) @ eps_x[t][..., None])[..., 0]
        slot = t % num_samples
        cs[slot], cm[slot], cc[slot] = samp, mean, cov
        x[missing_mask] = samp.flatten()
    return cs, cm, cc, accepted, x


def mixture_loglik(cond_mean: Tensor, cond_cov: Tensor, target: Tensor) -> Tensor:
    """Held-out log-likelihood under the equal-weight mixture of the stored conditional Gaussians
    (src/imputation_lib.py:230-231): logsumexp_k log N(target; mean_k, cov_k) − log K."""
    K = cond_mean.shape[0]
    return torch.logsumexp(mvn_log_prob(target, cond_mean, cond_cov), dim=0) - math.log(K)


def cast_state(sd: Dict[str, Tensor], dtype) -> Dict[str, Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
