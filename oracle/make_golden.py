"""Generate golden vectors by importing and running the reference itself (CPU, this container only).

Usage:  cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference (read-only at /root/reference) is imported unmodified; the one package it needs that
is not installed here (tensorboard) is replaced by an empty stand-in module before the import, which
is touched only inside ``fit()``.  Outputs are small ``.npz`` files of inputs and expected outputs
under ``tests/golden/``; no reference source is copied.  The GPU box never runs this script.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

stub = types.ModuleType("torch.utils.tensorboard")
stub.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = stub
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from src.vae_models import CVAE                      # noqa: E402
from src.conditioning_lib import Conditioner         # noqa: E402
from src import testing_lib, imputation_lib, utils as ref_utils   # noqa: E402

torch.autograd.set_detect_anomaly(False)
torch.set_num_threads(8)


def tiny_model_cfg(lik="dict-gauss", H=32, L=5, V=9, nh=2):
    post = {"dist_type": "normal", "learn_sigma": True, "sigma_lim": 0.05, "total_max_std": 3.0,
            "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": H, "num_hidden_layers": nh,
            "dropout": False, "dropout_rate": 0.2, "batch_normalization": True, "resnet": False}
    if lik == "dict-gauss":
        like = {"dist_type": "dict-gauss", "vocab_size": V, "learn_sigma": True, "sigma_lim": 1e-4,
                "marginal_std_lim": 0.1, "total_max_std": 6.0, "mu_upper_lim": 5.0, "mu_lower_lim": -3,
                "num_neurons": H, "num_hidden_layers": nh, "dropout": False, "dropout_rate": 0.2,
                "batch_normalization": True, "resnet": False}
    else:
        like = {"dist_type": "normal", "learn_sigma": True, "sigma_lim": 0.1, "mu_upper_lim": 5.0,
                "mu_lower_lim": -3, "num_neurons": H, "num_hidden_layers": nh, "dropout": False,
                "batch_normalization": True}
    return {"latent_dim": L, "distribution_dict": {"posterior": post, "likelihood": like}}


def make_conditioner(n, rng):
    months = rng.integers(1, 13, size=(n, 1))
    weekdays = rng.integers(0, 7, size=(n, 1))
    cset = {"months": months, "weekdays": weekdays}
    cond = Conditioner(tags=["months", "weekdays"], supports=[list(range(1, 13)), list(range(7))],
                       types=["circ", "circ"], condition_set=cset)
    return cond, cset


def synth_profiles(n, d, rng):
    x = np.clip(rng.normal(1.0, 1.0, size=(n, d)), -3, 5)
    x[rng.random((n, d)) < 0.1] = -3.0
    return x.astype(np.float32)


def np_state(sd):
    return {"sd::" + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


class Cdim:
    def __init__(self, c): self.cond_dim = c


class DropLog:
    """While active, ``torch.nn.Dropout`` modules draw their scaled mask with torch's own dropout applied to a
    tensor of ones, multiply the input by it, and the keep flags are recorded in call order."""
    def __init__(self):
        self.keep = []

    def __enter__(self):
        self._fwd = torch.nn.Dropout.forward
        log = self
        def forward(mod, x):
            if not mod.training or mod.p == 0.0:
                return x
            m = torch.nn.functional.dropout(torch.ones_like(x), mod.p, True)
            log.keep.append((m != 0).to(torch.uint8).numpy())
            return x * m
        torch.nn.Dropout.forward = forward
        return self

    def __exit__(self, *exc):
        torch.nn.Dropout.forward = self._fwd


def case_forward_train(name, lik, seed, D=6, C=4, B=8, S=3, three_d=False, steps=4, dropout=False, bn=True):
    rng = np.random.default_rng(seed)
    cfg = tiny_model_cfg(lik)
    if dropout:
        for d in cfg["distribution_dict"].values():
            d["dropout"] = True
    if not bn:
        for d in cfg["distribution_dict"].values():
            d["batch_normalization"] = False
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    L = cfg["latent_dim"]
    out = {"model_cfg": json.dumps(cfg), "D": D, "C": C, "B": B, "S": S}
    out.update(np_state(model.state_dict()))
    x = torch.tensor(synth_profiles(B, D, rng))
    if three_d:
        c = torch.tensor(rng.normal(size=(B, S, C)).astype(np.float32))
    else:
        c = torch.tensor(rng.normal(size=(B, C)).astype(np.float32))
    out["x"], out["c"] = x.numpy(), c.numpy()

    # forward + loss in train and eval mode with captured eps
    for mode in ("train", "eval"):
        # fresh copy so train-mode BN updates do not leak between the two
        torch.manual_seed(seed)
        m2 = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
        m2.load_state_dict(model.state_dict())
        m2.train() if mode == "train" else m2.eval()
        torch.manual_seed(seed + 100)
        eps = torch.randn((S, B, L)) if not three_d else torch.randn((1, S, B, L))
        torch.manual_seed(seed + 100)
        with DropLog() as dl, DrawLog() as rl:
            xd, zd = m2.forward(x, c, num_mc_samples=S)
        if dropout:      # dropout draws come first in the stream, so take the noise the forward really used
            eps = rl.randn[0]
        if mode == "train":
            for i, k in enumerate(dl.keep):
                out[f"train_keep_{i}"] = k
        loss = m2.loss(x, xd["params"], zd["params"], beta=0.7, prior_params=m2.prior_params)
        out[f"{mode}_eps"] = eps.reshape(S, B, L).numpy()
        out[f"{mode}_lik_mu"] = xd["params"]["mu"].detach().numpy()
        out[f"{mode}_lik_sigma"] = xd["params"]["sigma"].detach().numpy()
        out[f"{mode}_post_mu"] = zd["params"]["mu"].detach().numpy()
        out[f"{mode}_post_sigma"] = zd["params"]["sigma"].detach().numpy()
        out[f"{mode}_z"] = zd["samples"].detach().numpy()
        out[f"{mode}_rll_rows"] = m2.reconstruction_loglikelihood(x, xd["params"]).detach().numpy()
        for k in ("loss", "elbo", "rll", "kl"):
            out[f"{mode}_{k}"] = np.float64(loss[k].item())
        if mode == "train":
            loss["loss"].backward()
            for k, p in m2.named_parameters():
                out["grad::" + k] = p.grad.detach().numpy()
            out.update({"after_fwd_" + k: v for k, v in np_state(m2.state_dict()).items() if "running" in k or "tracked" in k})
        if lik == "dict-gauss" and mode == "eval":
            out["eval_cov"] = m2.decoder.create_covariance_matrix(xd["params"]).detach().numpy()
            torch.manual_seed(seed + 7)
            e2 = torch.randn((2, *xd["params"]["sigma"].shape))
            torch.manual_seed(seed + 7)
            out["eval_dict_rsample_eps"] = e2.numpy()
            out["eval_dict_rsample"] = m2.decoder.rsample(xd["params"], num_samples=2).detach().numpy()

    # train_core steps (2-D conditions only: that is what training uses)
    if not three_d:
        torch.manual_seed(seed)
        m3 = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
        m3.load_state_dict(model.state_dict())
        m3.train()
        m3.train_kwargs = {"num_mc_samples": S, "beta": 0.7, "gradient_clipping": True,
                           "gradient_clipping_kwargs": {"max_norm": 1.0}}
        optim = m3.get_optimizer(lr=1e-3, weight_decay=1e-5)
        xs, cs, es, ls = [], [], [], []
        for t in range(steps):
            xb = torch.tensor(synth_profiles(B, D, rng))
            cb = torch.tensor(rng.normal(size=(B, C)).astype(np.float32))
            torch.manual_seed(seed + 1000 + t)
            eps = torch.randn((S, B, L))
            torch.manual_seed(seed + 1000 + t)
            with DropLog() as dl, DrawLog() as rl:
                loss = m3.train_core((xb, cb), optim)
            if dropout:
                eps = rl.randn[0]
            for i, k in enumerate(dl.keep):
                out[f"step{t}_keep_{i}"] = k
            xs.append(xb.numpy()); cs.append(cb.numpy()); es.append(eps.numpy())
            ls.append([loss[k].item() for k in ("loss", "elbo", "rll", "kl")])
        out["steps_x"], out["steps_c"], out["steps_eps"] = np.stack(xs), np.stack(cs), np.stack(es)
        out["steps_losses"] = np.array(ls, dtype=np.float64)
        out.update({"final::" + k: v.detach().numpy() for k, v in m3.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def case_config0_shape(name="config0_smallbatch", seed=2112, B=64, S=4, steps=3):
    """config0 architecture (H=1000, D=48, L=48, V=100) at a small batch; weights are not stored
    (41 MB): the fixture pins seed-reproducible construction plus the per-step loss terms and a few
    parameter slices after the last step."""
    cfg = json.load(open(os.path.join(REF, "config_files", "config0.json")))
    mcfg = cfg["model"]
    D, C, L = 48, 4, mcfg["latent_dim"]
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=Cdim(C), **mcfg)
    model.train()
    model.train_kwargs = {"num_mc_samples": S, "beta": cfg["train"]["beta"], "gradient_clipping": True,
                          "gradient_clipping_kwargs": cfg["train"]["gradient_clipping_kwargs"]}
    out = {"model_cfg": json.dumps(mcfg), "seed": seed, "D": D, "C": C, "B": B, "S": S,
           "lr": cfg["train"]["lr"], "weight_decay": cfg["train"]["weight_decay"], "beta": cfg["train"]["beta"]}
    sd0 = model.state_dict()
    for k in ("encoder.parameterizer.block_dict.input.input_layer.0.weight",
              "decoder.parameterizer.block_dict.sigma.output_layer.0.bias", "decoder.SigmaMapper"):
        out["init::" + k] = sd0[k].detach().numpy().reshape(-1)[:64].copy()
    optim = model.get_optimizer(lr=cfg["train"]["lr"], weight_decay=cfg["train"]["weight_decay"])
    xs, cs, es, ls = [], [], [], []
    for t in range(steps):
        xb = torch.tensor(synth_profiles(B, D, rng))
        ang = rng.uniform(0, 2 * np.pi, size=(B, 2))
        cb = torch.tensor(np.concatenate([np.cos(ang), np.sin(ang)], 1).astype(np.float32))
        torch.manual_seed(seed + 1000 + t)
        eps = torch.randn((S, B, L))
        torch.manual_seed(seed + 1000 + t)
        loss = model.train_core((xb, cb), optim)
        xs.append(xb.numpy()); cs.append(cb.numpy()); es.append(eps.numpy())
        ls.append([loss[k].item() for k in ("loss", "elbo", "rll", "kl")])
    out["steps_x"], out["steps_c"], out["steps_eps"] = np.stack(xs), np.stack(cs), np.stack(es)
    out["steps_losses"] = np.array(ls, dtype=np.float64)
    sd1 = model.state_dict()
    for k in ("encoder.parameterizer.block_dict.input.input_layer.0.weight",
              "decoder.parameterizer.block_dict.input.middle_layers.1.0.weight",
              "decoder.parameterizer.block_dict.sigma.output_layer.0.bias",
              "decoder.parameterizer.block_dict.mu.input_layer.2.running_var", "decoder.SigmaMapper"):
        out["final::" + k] = sd1[k].detach().numpy().reshape(-1)[:64].copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, ls)


class DrawLog:
    """Records every torch.randn / torch.randn_like / torch.rand result while active."""
    def __init__(self):
        self.randn, self.randn_like, self.rand = [], [], []

    def __enter__(self):
        self._r, self._rl, self._u = torch.randn, torch.randn_like, torch.rand
        def randn(*a, **k):
            t = self._r(*a, **k); self.randn.append(t.clone()); return t
        def randn_like(*a, **k):
            t = self._rl(*a, **k); self.randn_like.append(t.clone()); return t
        def rand(*a, **k):
            t = self._u(*a, **k); self.rand.append(t.clone()); return t
        torch.randn, torch.randn_like, torch.rand = randn, randn_like, rand
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.rand = self._r, self._rl, self._u


def case_scoring(name="scoring_tiny", seed=5, N=20, K=7, D=6):
    rng = np.random.default_rng(seed)
    cfg = tiny_model_cfg("dict-gauss")
    cond, cset = make_conditioner(N, rng)
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=cond, **cfg)
    # make BN running stats non-trivial, then eval
    model.train()
    with torch.no_grad():
        for _ in range(3):
            model.forward(torch.tensor(synth_profiles(16, D, rng)), torch.tensor(rng.normal(size=(16, 4)).astype(np.float32)), num_mc_samples=2)
    model.eval()
    X = synth_profiles(N, D, rng)
    with DrawLog() as log:
        ll, rlls, kl = testing_lib.mass_loglikelihood(model, X, cond, cset, num_mc_samples=K, batch_size=8, mc_sample_batch_size=3, device="cpu")
    # first randn call is the shape probe (src/testing_lib.py:27); the rest are [chunk, batch, L] draws in (batch, chunk) order
    draws = log.randn[1:]
    L = cfg["latent_dim"]
    eps = np.zeros((K, N, L), np.float32)
    bi = 0
    it = iter(draws)
    for k0 in range(0, N, 8):
        for i0 in range(0, K, 3):
            d = next(it).numpy()
            eps[i0:i0 + d.shape[0], k0:k0 + d.shape[1]] = d
    out = {"model_cfg": json.dumps(cfg), "D": D, "N": N, "K": K, "x": X,
           "c": cond.transform(cset).astype(np.float32), "months": cset["months"], "weekdays": cset["weekdays"],
           "eps": eps, "ll": ll, "rlls": rlls, "kl": kl}
    out.update(np_state(model.state_dict()))
    # prior-sample scoring (src/vae_models.py:283-290 + decoder.log_likelihood)
    ctens = torch.tensor(cond.transform(cset)).float()
    with DrawLog() as log2:
        sdict = model.sample(ctens, num_samples_prior=4, num_samples_likelihood=1)
    z = log2.randn[0].reshape(4, N, L)
    out["prior_z"] = z.numpy()
    out["prior_rll"] = model.decoder.log_likelihood(torch.tensor(X), sdict["params"]).detach().numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def case_test_metrics(name="test_metrics_tiny", seed=12, N=18, K=6, D=6):
    """The per-profile metrics of scripts/collect_testing_results.py:84-113 on a tiny model, with every draw logged."""
    rng = np.random.default_rng(seed)
    cfg = tiny_model_cfg("dict-gauss")
    cond, cset = make_conditioner(N, rng)
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=cond, **cfg)
    model.train()
    with torch.no_grad():
        for _ in range(3):
            model.forward(torch.tensor(synth_profiles(16, D, rng)), torch.tensor(rng.normal(size=(16, 4)).astype(np.float32)), num_mc_samples=2)
    model.eval()
    X = synth_profiles(N, D, rng)
    with DrawLog() as log:
        x_rec, z_rec, rlls = testing_lib.mass_reconstruction(model, X, cond, cset, num_mc_samples=K, batch_size=7, mc_sample_batch_size=4, device="cpu")
    L = cfg["latent_dim"]
    eps = np.zeros((K, N, L), np.float32)
    it = iter(log.randn[1:])                      # first call is the shape probe (src/testing_lib.py:27)
    for k0 in range(0, N, 7):
        for i0 in range(0, K, 4):
            d = next(it).numpy()
            eps[i0:i0 + d.shape[0], k0:k0 + d.shape[1]] = d
    with torch.no_grad():
        post_ll = model.encoder.log_likelihood(z_rec["samples"], z_rec["params"])
        prior_ll = model.encoder.log_likelihood(z_rec["samples"], model.prior_params)
        kl = model.kl_divergence(z_rec["params"], prior_params=model.prior_params)
    ll = (-torch.log(torch.tensor(rlls.shape[0])) + torch.logsumexp(-post_ll + rlls + prior_ll, dim=0)).numpy()
    with DrawLog() as log2:
        x_samples = model.decoder.sample(x_rec["params"])[0].cpu().detach().numpy()
    eps_x = log2.randn[0].numpy()[0]              # [K, N, V]
    errors = {f"l_{p}": np.mean((np.abs(x_samples - X[None, ...]) ** p).sum(axis=-1), axis=0) ** (1 / p) for p in (1, 2)}
    out = {"model_cfg": json.dumps(cfg), "D": D, "N": N, "K": K, "x": X, "c": cond.transform(cset).astype(np.float32),
           "months": cset["months"], "weekdays": cset["weekdays"], "eps": eps, "eps_x": eps_x,
           "loglikelihood": ll, "kl": kl.numpy(), "l_1": errors["l_1"], "l_2": errors["l_2"]}
    out.update(np_state(model.state_dict()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def case_imputation(name="imputation_tiny", seed=9, N=10, D=6, W=3, T=5, num_samples=3):
    rng = np.random.default_rng(seed)
    cfg = tiny_model_cfg("dict-gauss")
    cond, cset = make_conditioner(N, rng)
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=cond, **cfg)
    model.train()
    with torch.no_grad():
        for _ in range(3):
            model.forward(torch.tensor(synth_profiles(16, D, rng)), torch.tensor(rng.normal(size=(16, 4)).astype(np.float32)), num_mc_samples=2)
    model.eval()
    X = synth_profiles(N, D, rng)
    x_missing = X.copy()
    x_missing[:, -W:] = np.nan
    target = X[:, -W:]
    with DrawLog() as log:
        ll, samples = imputation_lib.mass_cvae_imputation_with_loglikelihood(
            model, x_missing, target, cond, cset, batch_size=6, num_samples=num_samples, num_iter=0,
            warmup_steps=T, verbose_freq=1000, device="cpu")
    # per batch: 1 slack randn + 1 slack randn_like (src/imputation_lib.py:66,70), then T (randn, randn_like) pairs
    L = cfg["latent_dim"]
    eps_z = np.zeros((T, N, L), np.float32)
    eps_x = np.zeros((T, N, W), np.float32)
    r, rl = iter(log.randn), iter(log.randn_like)
    for b0 in range(0, N, 6):
        next(r); next(rl)
        for t in range(T):
            dz = next(r).numpy(); dx = next(rl).numpy()
            eps_z[t, b0:b0 + dz.shape[1]] = dz[0]
            eps_x[t, b0:b0 + dx.shape[0]] = dx
    # single-batch call for the ring buffers themselves
    torch.manual_seed(seed + 1)
    with DrawLog() as log2:
        cs, cm, cc, x_rec, z_rec = imputation_lib.cvae_imputation(model, x_missing, cond, cset, num_samples=num_samples,
                                                                  num_iter=0, warmup_steps=T, device="cpu")
    ez2 = np.stack([t.numpy()[0] for t in log2.randn[1:]])
    ex2 = np.stack([t.numpy() for t in log2.randn_like[1:]])
    out = {"model_cfg": json.dumps(cfg), "D": D, "N": N, "W": W, "T": T, "num_samples": num_samples,
           "x_missing": x_missing, "target": target, "c": cond.transform(cset).astype(np.float32),
           "eps_z": eps_z, "eps_x": eps_x, "batch_size": 6, "loglik": ll, "samples": samples,
           "one_eps_z": ez2, "one_eps_x": ex2, "one_cond_samples": cs.numpy(), "one_cond_mean": cm.numpy(),
           "one_cond_cov": cc.numpy()}
    out.update(np_state(model.state_dict()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def case_imputation_mh(name="imputation_mh_tiny", seed=21, N=9, D=6, W=3, T_pg=2, T_mh=4, num_samples=3):
    """Pseudo-Gibbs warm-up followed by Metropolis-within-Gibbs sweeps (src/imputation_lib.py:100-156,158-189) in one
    batch, every draw logged in call order: normal draws by shape, the acceptance uniforms, and the acceptance
    decisions' margins so that a replay in fp32 on another device cannot flip one."""
    rng = np.random.default_rng(seed)
    cfg = tiny_model_cfg("dict-gauss")
    cond, cset = make_conditioner(N, rng)
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=cond, **cfg)
    model.train()
    with torch.no_grad():
        for _ in range(3):
            model.forward(torch.tensor(synth_profiles(16, D, rng)), torch.tensor(rng.normal(size=(16, 4)).astype(np.float32)), num_mc_samples=2)
    model.eval()
    X = synth_profiles(N, D, rng)
    x_missing = X.copy()
    x_missing[:, -W:] = np.nan
    rhos = []
    orig_min = torch.minimum
    def minimum(a, b):
        r = orig_min(a, b); rhos.append(r.clone()); return r
    torch.manual_seed(seed + 1)
    torch.minimum = minimum
    try:
        with DrawLog() as log:
            cs, cm, cc, _, _ = imputation_lib.cvae_imputation(model, x_missing, cond, cset, num_samples=num_samples,
                                                              num_iter=T_mh, warmup_steps=T_pg, device="cpu")
    finally:
        torch.minimum = orig_min
    u = np.stack([t.numpy() for t in log.rand])                       # [T_mh, N]
    rho = np.stack([t.numpy() for t in rhos])                         # [T_mh, N]
    margin = float(np.abs(u - rho).min())
    assert margin > 2e-3, f"an acceptance decision sits {margin} from its threshold: pick another seed"
    out = {"model_cfg": json.dumps(cfg), "D": D, "N": N, "W": W, "T_pg": T_pg, "T_mh": T_mh, "num_samples": num_samples,
           "x_missing": x_missing, "c": cond.transform(cset).astype(np.float32),
           "randn_shapes": json.dumps([list(t.shape) for t in log.randn]),
           "randn_like_shapes": json.dumps([list(t.shape) for t in log.randn_like]),
           "accept_u": u, "accept_rho": rho, "accepted": (u < rho),
           "cond_samples": cs.numpy(), "cond_mean": cm.numpy(), "cond_cov": cc.numpy()}
    for i, t in enumerate(log.randn):
        out[f"randn_{i}"] = t.numpy()
    for i, t in enumerate(log.randn_like):
        out[f"randn_like_{i}"] = t.numpy()
    out.update(np_state(model.state_dict()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "randn", [list(t.shape) for t in log.randn], "randn_like", [list(t.shape) for t in log.randn_like],
          "accept rate", float((u < rho).mean()), "margin", margin)


class ListLoader:
    """A loader that yields fixed [x, c] batches; ``dataset`` only has to have the right length
    (src/vae_models.py:196,331,336 iterate the loader and read ``len(loader.dataset)``)."""
    def __init__(self, batches, n):
        self.batches, self.dataset = batches, range(n)
        self.batch_sampler = types.SimpleNamespace(batch_size=batches[0][0].shape[0])     # read for the progress bar only (:176)
        self.drop_last = True

    def __iter__(self):
        return iter([[x, c] for x, c in self.batches])


def _warm_bn(model, D, rng, calls=3):
    model.train()
    with torch.no_grad():
        for _ in range(calls):
            model.forward(torch.tensor(synth_profiles(16, D, rng)), torch.tensor(rng.normal(size=(16, 4)).astype(np.float32)), num_mc_samples=2)


def case_validate(name="validate_tiny", seed=31, D=6, C=4, S=3, sizes=(8, 8, 4)):
    """``CVAE.validate`` (src/vae_models.py:327-337) over three ragged batches, once with 2-D conditions and once with
    sampled (3-D) conditions ``[B,S,C]``; every latent draw is logged in call order."""
    rng = np.random.default_rng(seed)
    cfg = tiny_model_cfg("dict-gauss")
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    _warm_bn(model, D, rng)
    out = {"model_cfg": json.dumps(cfg), "D": D, "C": C, "S": S, "sizes": np.array(sizes)}
    out.update(np_state(model.state_dict()))
    n = sum(sizes)
    for tag, three_d in (("2d", False), ("3d", True)):
        batches = []
        for i, b in enumerate(sizes):
            x = torch.tensor(synth_profiles(b, D, rng))
            c = torch.tensor(rng.normal(size=(b, S, C) if three_d else (b, C)).astype(np.float32))
            batches.append((x, c))
            out[f"{tag}_x{i}"], out[f"{tag}_c{i}"] = x.numpy(), c.numpy()
        torch.manual_seed(seed + 5)
        with DrawLog() as log:
            val = model.validate(ListLoader(batches, n), num_mc_samples=S, beta=0.7, device="cpu")
        assert len(log.randn) == len(sizes)
        for i, e in enumerate(log.randn):
            out[f"{tag}_eps{i}"] = e.numpy()
        out[f"{tag}_val"] = np.array([val["elbo"], val["rll"], val["kl"]], dtype=np.float64)
        assert not model.training
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, out["2d_val"], out["3d_val"])


def case_fit(name="fit_tiny", seed=41, D=6, C=4, S=2, B=8, n_train=32, val_sizes=(8, 4), epochs=6):
    """A short ``fit`` run of the reference (src/vae_models.py:126-238) on fixed batches: validation every 2 iterations
    drives ReduceLROnPlateau(mode=max, threshold_mode=abs) and EarlyStopping (src/utils.py:156-180); the stop takes
    effect at the end of the epoch it was raised in (:235).  Logged: every latent draw in call order, the loss terms of
    every iteration, every validation result, the learning rate after every validation, the number of iterations run,
    BatchNorm counters and the final state."""
    import tempfile
    rng = np.random.default_rng(seed)
    cfg = tiny_model_cfg("dict-gauss")
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    out = {"model_cfg": json.dumps(cfg), "D": D, "C": C, "S": S, "B": B}
    out.update(np_state(model.state_dict()))
    train = [(torch.tensor(synth_profiles(B, D, rng)), torch.tensor(rng.normal(size=(B, C)).astype(np.float32))) for _ in range(n_train // B)]
    val = [(torch.tensor(synth_profiles(b, D, rng)), torch.tensor(rng.normal(size=(b, C)).astype(np.float32))) for b in val_sizes]
    for i, (x, c) in enumerate(train):
        out[f"train_x{i}"], out[f"train_c{i}"] = x.numpy(), c.numpy()
    for i, (x, c) in enumerate(val):
        out[f"val_x{i}"], out[f"val_c{i}"] = x.numpy(), c.numpy()
    fit_kwargs = dict(lr=1e-3, weight_decay=1e-5, gradient_clipping=True, gradient_clipping_kwargs={"max_norm": 1.0},
                      lr_scheduling=True, lr_scheduling_kwargs={"threshold": 15.0, "factor": 0.2, "patience": 1, "min_lr": 1e-5},
                      beta=1.0, num_mc_samples=S, epochs=epochs, verbose_freq=1000, tensorboard=False, validation_freq=2,
                      validation_mc_samples=3, device="cpu", earlystopping=True, earlystopping_kwargs={"patience": 2, "delta": 30.0},
                      save_epoch_freq=None)
    out["fit_kwargs"] = json.dumps(fit_kwargs)
    train_log, val_log, lr_log = [], [], []
    tc, vd = model.train_core, model.validate
    holder = {}
    def train_core(inputs, optim, **k):
        holder["optim"] = optim
        loss = tc(inputs, optim, **k)
        train_log.append([loss[q].item() for q in ("loss", "elbo", "rll", "kl")])
        return loss
    def validate(*a, **k):
        r = vd(*a, **k)
        val_log.append([r["elbo"], r["rll"], r["kl"], len(train_log)])
        return r
    model.train_core, model.validate = train_core, validate
    step_orig = torch.optim.lr_scheduler.ReduceLROnPlateau.step
    def sched_step(self_, metrics, *a, **k):
        r = step_orig(self_, metrics, *a, **k)
        lr_log.append(self_.optimizer.param_groups[0]["lr"])
        return r
    torch.optim.lr_scheduler.ReduceLROnPlateau.step = sched_step

    class W:
        log_dir = tempfile.mkdtemp()

    class Bar:                                      # the early-stopping call needs a progress bar object (:213)
        def __init__(self, **k): pass
        def update(self, n): pass
        def reset(self): pass
        def write(self, msg): print(msg)
    torch.manual_seed(seed + 9)
    try:
        with DrawLog() as log:
            model.fit(ListLoader(train, n_train), ListLoader(val, sum(val_sizes)), tqdm_func=Bar, writer=W(), **fit_kwargs)
    finally:
        torch.optim.lr_scheduler.ReduceLROnPlateau.step = step_orig
    del model.train_core, model.validate
    out["n_draws"] = len(log.randn)
    for i, e in enumerate(log.randn):
        out[f"eps_{i}"] = e.numpy()
    out["train_losses"] = np.array(train_log, dtype=np.float64)
    out["val_log"] = np.array(val_log, dtype=np.float64)
    out["lr_log"] = np.array(lr_log, dtype=np.float64)
    out["iterations"] = len(train_log)
    out["training_after_fit"] = int(model.training)
    out.update({"final::" + k: v.detach().numpy() for k, v in model.state_dict().items()})
    # the control decisions must not sit on a knife edge: replaying them in fp32 on another device may not flip one
    elbos = out["val_log"][:, 0]
    margins = []
    for thr in (fit_kwargs["lr_scheduling_kwargs"]["threshold"], fit_kwargs["earlystopping_kwargs"]["delta"]):
        best = elbos[0]
        for e in elbos[1:]:
            margins.append(abs(e - (best + thr)))
            if e > best + thr:
                best = e
    assert min(margins) > 0.05, f"a plateau / early-stopping decision sits {min(margins)} from its threshold: pick another seed"
    assert len(set(lr_log)) > 1, "the run never changed the learning rate: lengthen it"
    assert out["iterations"] < epochs * (n_train // B), "early stopping never fired: lengthen the run"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "iterations", out["iterations"], "val", out["val_log"].tolist(), "lr", lr_log, "margin", min(margins))


def case_config0_sgd(name="config0_sgd_steps", seed=2112, B=64, S=4, steps=4, lr=0.1):
    """config0 architecture, four ``train_core`` steps of the reference driven by plain SGD (``train_core`` takes whatever
    optimiser it is given, src/vae_models.py:317-325) on gradients clipped to norm 1: every step moves the weights 0.1
    along the gradient, so the loss terms of step t+1 test the whole backward pass of step t, and -- unlike Adam, whose
    first updates are sign-like and turn round-off-sized gradients into +-lr moves -- the update is linear in the
    gradient, so two fp32 implementations stay 1e-6 apart along the trajectory.  Stored per step: inputs, noise, loss
    terms, the first 64 values of every clipped gradient; after the last step (sum, sum of squares) and the first 64
    values of every state tensor."""
    cfg = json.load(open(os.path.join(REF, "config_files", "config0.json")))
    mcfg = cfg["model"]
    D, C, L = 48, 4, mcfg["latent_dim"]
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = CVAE(input_dim=D, conditioner=Cdim(C), **mcfg)
    model.train()
    model.train_kwargs = {"num_mc_samples": S, "beta": cfg["train"]["beta"], "gradient_clipping": True,
                          "gradient_clipping_kwargs": cfg["train"]["gradient_clipping_kwargs"]}
    out = {"model_cfg": json.dumps(mcfg), "seed": seed, "D": D, "C": C, "B": B, "S": S, "lr": lr, "beta": cfg["train"]["beta"]}
    optim = torch.optim.SGD(model.parameters(), lr=lr)
    xs, cs, es, ls = [], [], [], []
    names = [k for k, _ in model.named_parameters()]
    for t in range(steps):
        xb = torch.tensor(synth_profiles(B, D, rng))
        ang = rng.uniform(0, 2 * np.pi, size=(B, 2))
        cb = torch.tensor(np.concatenate([np.cos(ang), np.sin(ang)], 1).astype(np.float32))
        torch.manual_seed(seed + 1000 + t)
        eps = torch.randn((S, B, L))
        torch.manual_seed(seed + 1000 + t)
        loss = model.train_core((xb, cb), optim)
        out[f"grad{t}"] = np.stack([np.resize(p.grad.detach().numpy().reshape(-1)[:64], 64) for _, p in model.named_parameters()])
        xs.append(xb.numpy()); cs.append(cb.numpy()); es.append(eps.numpy())
        ls.append([loss[k].item() for k in ("loss", "elbo", "rll", "kl")])
        print("step", t, ls[-1], flush=True)
    out["grad_keys"] = json.dumps(names)
    out["steps_x"], out["steps_c"], out["steps_eps"] = np.stack(xs), np.stack(cs), np.stack(es)
    out["steps_losses"] = np.array(ls, dtype=np.float64)
    sd = model.state_dict()
    keys = sorted(sd)
    out["final_keys"] = json.dumps(keys)
    out["final_sums"] = np.array([[float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in keys])
    out["final_head"] = np.stack([np.resize(sd[k].detach().double().numpy().reshape(-1)[:64], 64) for k in keys])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def case_pointwise(name="pointwise"):
    xs = torch.tensor([-30.0, -10.0, -1.0, 0.0, 1.0, 4.9, 5.0, 5.0001, 5.1, 10.0, 40.0])
    out = {"softplus_in": xs.numpy(), "softplus_out": ref_utils.to_sigma(xs).numpy()}
    mu = torch.tensor([[0.3, -1.2, 2.0]]); sg = torch.tensor([[0.5, 1.0, 0.05]]); t = torch.tensor([[0.0, 1.0, 2.1]])
    out["lp_mu"], out["lp_sigma"], out["lp_t"] = mu.numpy(), sg.numpy(), t.numpy()
    out["lp_out"] = ref_utils.log_prob("Normal", {"mu": mu, "sigma": sg}, t).numpy()
    out["kl_out"] = ref_utils.kl_divergence("Normal", {"mu": mu, "sigma": sg}, {"mu": torch.zeros(3), "sigma": torch.ones(3)}).numpy()
    U = torch.tensor(np.random.default_rng(3).normal(size=(4, 6)).astype(np.float32))
    out["U"], out["U_norm"] = U.numpy(), ref_utils.matrix_normalizer(U).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


def case_conditioner(name="conditioner"):
    rng = np.random.default_rng(11)
    n, T = 12, 5
    months = rng.integers(1, 13, size=(n, 1)); weekdays = rng.integers(0, 7, size=(n, 1))
    day_before = rng.normal(size=(n, 3)); gammas = rng.gamma(2.0, 1.0, size=(n, T)) + 0.01
    out = {"months": months, "weekdays": weekdays, "day_befores": day_before, "users": gammas}
    for style in ("mean", "embed", "scaled", "sample"):
        cset = {"months": months, "weekdays": weekdays, "day_befores": day_before, "users": gammas}
        cond = Conditioner(tags=["months", "weekdays", "day_befores", "users"],
                           supports=[list(range(1, 13)), list(range(7)), [-3, 3], [0.01, 10.0]],
                           types=["circ", "circ", "identity", "dir"], condition_set=cset)
        cond.transformers["users"].transform_style = style
        np.random.seed(4)
        res = cond.transform(cset, num_samples=3)
        out["out_" + style] = res
        out["cond_dim"] = cond.cond_dim
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "dropout":      # only the dropout case (leaves the other fixtures as they are)
        case_forward_train("tiny_dropout_2d", "dict-gauss", seed=6, B=16, dropout=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":      # only the test-metrics case
        case_test_metrics()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mh":           # only the Metropolis-within-Gibbs case
        case_imputation_mh()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "validate":     # only the validate case
        case_validate()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fit":          # only the fit-loop case
        case_fit()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sgd":          # only the config0 SGD-steps case
        case_config0_sgd()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "nobn":         # only the case without BatchNorm
        case_forward_train("tiny_nobn_2d", "dict-gauss", seed=8, B=16, dropout=True, bn=False)
        sys.exit(0)
    case_pointwise()
    case_conditioner()
    case_forward_train("tiny_dict_2d", "dict-gauss", seed=1)
    case_forward_train("tiny_dict_3d", "dict-gauss", seed=2, three_d=True)
    case_forward_train("tiny_normal_2d", "normal", seed=3)
    case_forward_train("tiny_dropout_2d", "dict-gauss", seed=6, B=16, dropout=True)
    case_forward_train("tiny_nobn_2d", "dict-gauss", seed=8, B=16, dropout=True, bn=False)
    case_scoring()
    case_test_metrics()
    case_imputation()
    case_imputation_mh()
    case_config0_shape()
    case_validate()
    case_fit()
    case_config0_sgd()
