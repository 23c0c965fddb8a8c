"""GPU: Monte-Carlo scoring and pseudo-Gibbs imputation against the reference's golden vectors (noise injected
by substituting ``torch.randn`` with the recorded draws), size-independent properties at larger shapes, and the
training entry point end to end on synthetic data."""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import golden_state, load_golden, rel_err

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def make_cond(g):
    from behavio_temporal_vae_amd.conditioning_lib import Conditioner
    cset = {"months": g["months"], "weekdays": g["weekdays"]}
    return Conditioner(tags=["months", "weekdays"], supports=[list(range(1, 13)), list(range(7))], types=["circ", "circ"], condition_set=cset), cset


class CondArray:
    """Conditioner stand-in whose transform returns pre-encoded rows (the imputation fixture stores c directly)."""
    types, tags, transformers = ["identity"], ["c"], {}

    def __init__(self, dim):
        self.cond_dim = dim

    def transform(self, cset, num_samples=1):
        return cset["c"]


def build(g, dev, cond):
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = json.loads(str(g["model_cfg"]))
    m = CVAE(input_dim=int(g["D"]), conditioner=cond, **cfg)
    m.load_state_dict(golden_state(g))
    return m.to(dev).eval()


class RandnQueue:
    """Replaces torch.randn: shapes listed in ``feeds`` pop recorded draws, anything else returns zeros."""

    def __init__(self, feeds):
        self.feeds = {tuple(k): list(v) for k, v in feeds.items()}

    def __enter__(self):
        self.orig = torch.randn

        def fake(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            dev = kw.get("device", "cpu")
            q = self.feeds.get(shape)
            if q:
                return q.pop(0).to(dev)
            return torch.zeros(shape, device=dev)
        torch.randn = fake
        return self

    def __exit__(self, *a):
        torch.randn = self.orig


def test_iw_scoring_vs_reference(dev):
    from behavio_temporal_vae_amd import testing_lib
    g = load_golden("scoring_tiny")
    cond, cset = make_cond(g)
    m = build(g, dev, cond)
    x, c, eps = T(g["x"]).to(dev), T(g["c"]).to(dev), T(g["eps"]).to(dev)
    ll, rlls, kl = testing_lib.iw_loglikelihood_device(m, x, c, int(g["K"]), mc_sample_batch_size=3, eps=eps, return_rlls=True)
    assert rel_err(rlls.cpu().numpy(), g["rlls"]) < 1e-4
    assert rel_err(ll.cpu().numpy(), g["ll"]) < 1e-4
    assert rel_err(kl.cpu().numpy(), g["kl"]) < 1e-4
    # driver with the reference's signature: same batches (8) and MC chunks (3) as the fixture, draws replayed in order
    K, N, L = int(g["K"]), int(g["N"]), 5
    feeds = {}
    for b0 in range(0, N, 8):
        for i0 in range(0, K, 3):
            e = T(g["eps"][i0:i0 + 3, b0:b0 + 8])
            feeds.setdefault(tuple(e.shape), []).append(e)
    with RandnQueue(feeds):
        ll2, rl2, kl2 = testing_lib.mass_loglikelihood(m, g["x"], cond, cset, num_mc_samples=K, batch_size=8, mc_sample_batch_size=3, device="cuda:0")
    assert rel_err(ll2, g["ll"]) < 1e-4 and rel_err(rl2, g["rlls"]) < 1e-4 and rel_err(kl2, g["kl"]) < 1e-4
    feeds = {}
    for b0 in range(0, N, 8):
        for i0 in range(0, K, 3):
            e = T(g["eps"][i0:i0 + 3, b0:b0 + 8])
            feeds.setdefault(tuple(e.shape), []).append(e)
    with RandnQueue(feeds):
        x_rec, z_rec, rl3 = testing_lib.mass_reconstruction(m, g["x"], cond, cset, num_mc_samples=K, batch_size=8, mc_sample_batch_size=3, device="cuda:0")
    assert rel_err(rl3.numpy(), g["rlls"]) < 1e-4
    assert x_rec["params"]["mu"].shape == (K, N, int(g["D"])) and x_rec["params"]["sigma"].shape == (K, N, 9)
    assert z_rec["samples"].shape == (K, N, L) and z_rec["params"]["mu"].shape == (N, L)
    # the model's own loglikelihood method (one chunk of K draws)
    m.to(dev)
    with RandnQueue({(K, N, L): [T(g["eps"])]}):
        ll4 = m.loglikelihood(x, c, num_mc_samples=K)
    assert rel_err(ll4.cpu().numpy(), g["ll"]) < 1e-4
    # prior-sample scoring (model.sample + decoder.log_likelihood)
    m.to(dev)
    z = T(g["prior_z"])
    with RandnQueue({(4 * N, L): [z.reshape(4 * N, L)]}):
        sd = m.sample(c, num_samples_prior=4, num_samples_likelihood=1)
    pr = m.decoder.log_likelihood(x, sd["params"])
    assert rel_err(pr.cpu().numpy(), g["prior_rll"]) < 1e-4
    assert sd["samples"].shape == (1, 4, N, int(g["D"]))


def test_iw_scoring_properties_large(dev):
    """K=1000-style streaming at a larger shape: chunking must not change the result, and K=1 reduces to
    rll - log q + log p(z) exactly."""
    from behavio_temporal_vae_amd import testing_lib
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = {"latent_dim": 24, "distribution_dict": {
        "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": 200,
                      "num_hidden_layers": 2, "dropout": True, "dropout_rate": 0.2, "batch_normalization": True},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "sigma_lim": 1e-6, "marginal_std_lim": 0.05, "mu_lower_lim": -3,
                       "num_neurons": 200, "num_hidden_layers": 2, "dropout": True, "dropout_rate": 0.2, "batch_normalization": True}}}

    class C4:
        cond_dim = 4
    torch.manual_seed(3)
    m = CVAE(input_dim=24, conditioner=C4(), **cfg).to(dev).eval()
    N, K = 257, 100
    gen = torch.Generator().manual_seed(4)
    x, c = torch.randn(N, 24, generator=gen).to(dev), torch.randn(N, 4, generator=gen).to(dev)
    eps = torch.randn(K, N, 24, generator=gen).to(dev)
    a, ra, _ = testing_lib.iw_loglikelihood_device(m, x, c, K, mc_sample_batch_size=100, eps=eps, return_rlls=True)
    b, rb, _ = testing_lib.iw_loglikelihood_device(m, x, c, K, mc_sample_batch_size=7, eps=eps, return_rlls=True)
    assert torch.allclose(ra, rb, rtol=1e-5, atol=1e-4)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-4)
    # the IW estimate is bounded by its best and its mean-of-logs sample weight
    l1, r1, _ = testing_lib.iw_loglikelihood_device(m, x, c, 1, eps=eps[:1], return_rlls=True)
    assert torch.isfinite(l1).all() and (a >= l1 - math.log(K) - 1e-3).all()


@pytest.mark.timeout(600)
def test_gipuzkoa_k1000_scoring_vs_oracle(dev):
    """The scoring benchmark's own shape -- config0_gipuzkoa model (D=L=24, C=4, H=1000, V=100, eval mode with
    non-trivial BatchNorm running statistics), K=1000 latent draws -- on 64 profiles against the fp64 CPU oracle with
    the same injected noise: per-draw log-likelihoods, the importance-weighted log-likelihood and KL at 1e-4, in the
    chunking the benchmark uses (25 draws per decoder pass)."""
    import contextlib, io
    from oracle import refimpl as R
    from behavio_temporal_vae_amd import testing_lib
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = {"latent_dim": 24, "distribution_dict": {
        "posterior": {"dist_type": "normal", "learn_sigma": True, "sigma_lim": 0.05, "average_max_std": 1.0, "mu_upper_lim": 5.0,
                      "mu_lower_lim": -5.0, "num_neurons": 1000, "num_hidden_layers": 2, "dropout": True, "dropout_rate": 0.2,
                      "batch_normalization": True, "resnet": False},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "learn_sigma": True, "sigma_lim": 1e-6, "marginal_std_lim": 0.05,
                       "average_max_std": 1.0, "mu_upper_lim": 5.0, "mu_lower_lim": -3, "num_neurons": 1000, "num_hidden_layers": 2,
                       "dropout": True, "dropout_rate": 0.2, "batch_normalization": True, "resnet": False}}}

    class C4:
        cond_dim = 4
    N, K, D, L = 64, 1000, 24, 24
    torch.manual_seed(2112)
    with contextlib.redirect_stdout(io.StringIO()):
        m = CVAE(input_dim=D, conditioner=C4(), **cfg)
    gen = torch.Generator().manual_seed(8)
    sd = m.state_dict()
    for k, v in sd.items():                       # running statistics as after some training, not (0, 1)
        if k.endswith("running_mean"):
            v.copy_(0.3 * torch.randn(v.shape, generator=gen))
        elif k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=gen))
    m.load_state_dict(sd)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(N, D, generator=gen).add_(1.0).clamp_(-3, 5)
    ang = torch.rand(N, 2, generator=gen) * (2 * math.pi)
    c = torch.cat([torch.cos(ang), torch.sin(ang)], 1)
    eps = torch.randn(K, N, L, generator=gen)
    spec = R.model_spec(D, 4, cfg)
    with torch.no_grad():
        ll64, rl64, kl64 = R.iw_loglikelihood(R.cast_state(sd, torch.float64), spec, x.double(), c.double(), eps.double())
    m = m.to(dev).eval()
    ll, rlls, kl = testing_lib.iw_loglikelihood_device(m, x.to(dev), c.to(dev), K, mc_sample_batch_size=25, eps=eps.to(dev), return_rlls=True)
    assert rel_err(rlls.cpu().numpy(), rl64.numpy()) < 1e-4
    assert rel_err(ll.cpu().numpy(), ll64.numpy()) < 1e-4
    assert rel_err(kl.cpu().numpy(), kl64.numpy()) < 1e-4
    # the same against the oracle with the hidden activations kept in the pair format (what the benchmark's 819,200-row
    # chunks do; these 1,600-row chunks are below the row count from which the engine switches on its own)
    from behavio_temporal_vae_amd import engine as E
    keep = E.EVAL_PAIRS_MIN_ROWS
    try:
        E.EVAL_PAIRS_MIN_ROWS = 0
        ll, rlls, kl = testing_lib.iw_loglikelihood_device(m, x.to(dev), c.to(dev), K, mc_sample_batch_size=25, eps=eps.to(dev), return_rlls=True)
    finally:
        E.EVAL_PAIRS_MIN_ROWS = keep
    assert rel_err(rlls.cpu().numpy(), rl64.numpy()) < 1e-4
    assert rel_err(ll.cpu().numpy(), ll64.numpy()) < 1e-4
    assert rel_err(kl.cpu().numpy(), kl64.numpy()) < 1e-4


def test_pseudo_gibbs_vs_reference(dev):
    from behavio_temporal_vae_amd import imputation_lib
    g = load_golden("imputation_tiny")
    cond = CondArray(4)
    m = build(g, dev, cond)
    D, W, Tn, ns, N, L = int(g["D"]), int(g["W"]), int(g["T"]), int(g["num_samples"]), int(g["N"]), 5
    xm = g["x_missing"].astype(np.float32)
    # single call over all rows, draws replayed in the order the reference made them
    feeds = {(1, N, L): [T(e)[None] for e in g["one_eps_z"]], (N, W): [T(e) for e in g["one_eps_x"]]}
    with RandnQueue(feeds):
        cs, cm, cc, x_rec, z_rec = imputation_lib.cvae_imputation(m, xm, cond, {"c": g["c"]}, num_samples=ns, num_iter=0, warmup_steps=Tn, device="cuda:0")
    assert np.allclose(cm.cpu().numpy(), g["one_cond_mean"], rtol=2e-4, atol=5e-5)
    assert np.allclose(cc.cpu().numpy(), g["one_cond_cov"], rtol=2e-4, atol=5e-6)
    assert np.allclose(cs.cpu().numpy(), g["one_cond_samples"], rtol=2e-4, atol=5e-5)
    assert x_rec["params"]["mu"].shape == (ns, N, D) and z_rec["samples"].shape == (ns, N, L)
    # batched driver with held-out log-likelihood (batch size 6 -> ragged last batch of 4)
    bs = int(g["batch_size"])
    feeds = {}
    for b0 in range(0, N, bs):
        nb = min(bs, N - b0)
        for t in range(Tn):
            feeds.setdefault((1, nb, L), []).append(T(g["eps_z"][t, b0:b0 + nb])[None])
            feeds.setdefault((nb, W), []).append(T(g["eps_x"][t, b0:b0 + nb]))
    with RandnQueue(feeds):
        ll, samples = imputation_lib.mass_cvae_imputation_with_loglikelihood(m, xm, g["target"], cond, {"c": g["c"]}, batch_size=bs, num_samples=ns,
                                                                             num_iter=0, warmup_steps=Tn, device="cuda:0")
    assert np.allclose(samples, g["samples"], rtol=2e-4, atol=5e-5)
    assert rel_err(ll, g["loglik"]) < 2e-4
    assert imputation_lib.pseudo_gibbs.total_time > 0 and imputation_lib.sample_conditional.total_time > 0


def test_metropolis_within_gibbs_vs_reference(dev):
    """Pseudo-Gibbs warm-up + Metropolis-within-Gibbs sweeps (src/imputation_lib.py:100-156) with the reference's
    normal draws and acceptance uniforms replayed in call order: same accept decisions (the fixture's margins are
    > 2e-3, so fp32 round-off cannot flip one), same final conditional samples / means / covariances."""
    from behavio_temporal_vae_amd import imputation_lib
    g = load_golden("imputation_mh_tiny")
    cond = CondArray(4)
    m = build(g, dev, cond)
    N, W, L, ns = int(g["N"]), int(g["W"]), 5, int(g["num_samples"])
    Tpg, Tmh = int(g["T_pg"]), int(g["T_mh"])
    shapes = json.loads(str(g["randn_shapes"]))
    like_shapes = json.loads(str(g["randn_like_shapes"]))
    feeds = {(1, N, L): [T(g[f"randn_{i}"]) for i, sh in enumerate(shapes) if sh == [1, N, L]],
             (N, W): [T(g[f"randn_like_{i}"]) for i, sh in enumerate(like_shapes) if sh == [N, W]]}
    assert len(feeds[(1, N, L)]) == Tpg + 1 + Tmh and len(feeds[(N, W)]) == Tpg + Tmh
    us = [T(u) for u in g["accept_u"]]
    seen = []
    orig_rand, orig_where = torch.rand, torch.where

    def fake_rand(*size, **kw):
        return us.pop(0)

    def spy_where(*a, **k):
        r = orig_where(*a, **k)
        if len(a) == 1 and a[0].dtype == torch.bool and a[0].shape == (N,) and not a[0].is_cuda:
            seen.append(a[0].clone())
        return r

    torch.rand, torch.where = fake_rand, spy_where
    try:
        with RandnQueue(feeds) as rq:
            cs, cm, cc, x_rec, z_rec = imputation_lib.cvae_imputation(m, g["x_missing"].astype(np.float32), cond, {"c": g["c"]}, num_samples=ns,
                                                                      num_iter=Tmh, warmup_steps=Tpg, device="cuda:0")
    finally:
        torch.rand, torch.where = orig_rand, orig_where
    assert not us and not rq.feeds[(1, N, L)] and not rq.feeds[(N, W)]    # every recorded draw was consumed
    assert len(seen) == Tmh and np.array_equal(torch.stack(seen).numpy(), g["accepted"])
    assert x_rec is None and z_rec is None
    assert np.allclose(cm.cpu().numpy(), g["cond_mean"], rtol=2e-4, atol=5e-5)
    assert np.allclose(cc.cpu().numpy(), g["cond_cov"], rtol=2e-4, atol=5e-6)
    assert np.allclose(cs.cpu().numpy(), g["cond_samples"], rtol=2e-4, atol=5e-5)


def test_normal_decoder_covariance_helpers_and_conditional(dev):
    """GaussianNN as the decoder (src/submodels.py:120-124): ``create_covariance_matrix`` is diag(sigma^2),
    ``get_marginal_sigmas`` is sigma, and the imputation conditional equals the general Gaussian conditioning formula
    (src/imputation_lib.py:28-35) applied to that diagonal covariance -- mean mu_u, covariance diag(sigma_u^2) -- for a
    non-suffix window; a short pseudo-Gibbs chain runs on it and leaves the observed entries untouched."""
    from behavio_temporal_vae_amd import imputation_lib
    g = load_golden("tiny_normal_2d")
    cond = CondArray(4)
    m = build(g, dev, cond)
    x, c = T(g["x"]).to(dev), T(g["c"]).to(dev)
    D = int(g["D"])
    with torch.no_grad():
        xd, _ = m.forward(x, c, num_mc_samples=2)
    params = xd["params"]
    cov = m.decoder.create_covariance_matrix(params)
    assert cov.shape == (*params["sigma"].shape, D)
    assert torch.equal(cov, torch.diag_embed(params["sigma"] ** 2))
    assert torch.equal(m.decoder.get_marginal_sigmas(params), params["sigma"])
    obs = [0, 2, 5]
    un = [1, 3, 4]
    p0 = {k: v[0] for k, v in params.items()}
    mean, cc = imputation_lib.get_conditional_params(m.decoder, p0, x[:, obs], torch.tensor(obs))
    S64 = torch.diag_embed(p0["sigma"].double() ** 2)
    Soo, Suo, Suu = S64[:, obs][:, :, obs], S64[:, un][:, :, obs], S64[:, un][:, :, un]
    A = Suo @ torch.linalg.inv(Soo)
    ref_mean = p0["mu"].double()[:, un] + (A @ (x.double()[:, obs] - p0["mu"].double()[:, obs])[..., None])[..., 0]
    ref_cov = Suu - A @ Suo.transpose(-1, -2)
    assert torch.allclose(mean.double(), ref_mean, rtol=1e-6, atol=1e-7)
    assert torch.allclose(cc.double(), ref_cov, rtol=1e-6, atol=1e-9)
    xm = g["x"].astype(np.float32).copy()
    xm[:, un] = np.nan
    cs, cm, cv, x_rec, z_rec = imputation_lib.cvae_imputation(m, xm, cond, {"c": g["c"]}, num_samples=2, num_iter=0, warmup_steps=3, device="cuda:0")
    assert cs.shape == (2, xm.shape[0], 3) and cv.shape == (2, xm.shape[0], 3, 3) and torch.isfinite(cs).all()
    assert x_rec["params"]["sigma"].shape == (2, xm.shape[0], D)
    full = xm.copy()
    with pytest.raises(ValueError, match="nothing to impute"):
        imputation_lib.cvae_imputation(m, np.nan_to_num(full), cond, {"c": g["c"]}, num_samples=2, num_iter=0, warmup_steps=1, device="cuda:0")


def test_imputation_mask_semantics(dev):
    """Bit-exact index/mask handling: observed indices from row 0 only, row-major masked write-back, a mask whose
    count does not match raises like the reference's indexed assignment would."""
    from behavio_temporal_vae_amd import imputation_lib
    g = load_golden("imputation_tiny")
    cond = CondArray(4)
    m = build(g, dev, cond)
    N, D = 7, int(g["D"])
    x = np.random.default_rng(0).normal(size=(N, D)).astype(np.float32)
    xm = x.copy()
    xm[:, [1, 4]] = np.nan                       # a non-suffix window
    cs, cm, cc, _, _ = imputation_lib.cvae_imputation(m, xm, cond, {"c": g["c"][:N]}, num_samples=2, num_iter=0, warmup_steps=3, device="cuda:0")
    assert cs.shape == (2, N, 2) and cc.shape == (2, N, 2, 2)
    assert torch.isfinite(cs).all()
    # observed entries are never touched by the chain: rerun pseudo_gibbs directly and compare x
    m.to(dev)
    xt = torch.tensor(xm).to(dev)
    mask = torch.isnan(xt)
    xt[mask] = 0.0
    before = xt.clone()
    imputation_lib.pseudo_gibbs(m, xt, mask, cond, {"c": g["c"][:N]}, num_samples=2, num_iter=2, device="cuda:0")
    assert torch.equal(xt[~mask], before[~mask]) and not torch.equal(xt[mask], before[mask])
    bad = xm.copy()
    bad[3, 2] = np.nan                           # row 3 misses one more entry than row 0 says
    with pytest.raises(RuntimeError):
        imputation_lib.cvae_imputation(m, bad, cond, {"c": g["c"][:N]}, num_samples=2, num_iter=0, warmup_steps=1, device="cuda:0")


def test_training_entry_point_synthetic(dev, tmp_path):
    """config0-shaped plumbing run: a few iterations of ``main(config)`` on synthetic data with validation,
    LR scheduling, early stopping bookkeeping and checkpoint round trip through the reference's key names."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("training_entry", os.path.join(ROOT, "scripts", "training.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    cfg = tr.load_config(os.path.join(ROOT, "config_files", "config0.json"))
    for k in ("posterior", "likelihood"):
        cfg["model"]["distribution_dict"][k]["num_neurons"] = 96
    cfg["train"].update({"epochs": 2, "batch_size": 256, "validation_freq": 4, "verbose_freq": 3, "num_mc_samples": 4,
                         "validation_mc_samples": 2, "device": "cuda:0", "save_epoch_freq": 1})
    cfg["save_dir"] = str(tmp_path)
    model = tr.main(cfg, n_train=2048, n_val=512)
    run = model.log_dir
    for f in ("trained_model.pt", "config.json", "model_kwargs.json", "train_kwargs.json", "conditioner.pkl"):
        assert os.path.exists(os.path.join(run, f)), f
    sd = torch.load(os.path.join(run, "trained_model.pt"), weights_only=True)
    assert "decoder.SigmaMapper" in sd and "encoder.parameterizer.block_dict.input.middle_layers.1.2.running_var" in sd
    assert int(sd["encoder.parameterizer.block_dict.mu.input_layer.2.num_batches_tracked"]) == 16     # 2 epochs x 8 iterations
    from behavio_temporal_vae_amd.vae_models import CVAE

    class C4:
        cond_dim = 4
    m2 = CVAE(input_dim=48, conditioner=C4(), **cfg["model"])
    m2.log_dir = run
    m2.load()
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in sd.items())
    mk = json.load(open(os.path.join(run, "model_kwargs.json")))
    assert set(mk) == {"latent_dim", "distribution_dict", "_"}


def test_long_training_run_replayed_equals_eager(dev, tmp_path, monkeypatch):
    """A thousand iterations of ``main(config0)`` at its shipped size, with validation passes in between, once with the
    training step replayed from its HIP graph and once eager: the logged ELBO and KL agree all the way (a replay that
    keeps anything from an earlier replay -- operand maxima in the two-piece GEMM mode above all -- degrades training
    only after several hundred iterations, which the few-step tests cannot see), and the posterior has not collapsed."""
    import importlib.util
    import os
    from conftest import ROOT
    from behavio_temporal_vae_amd import vae_models as VM
    spec = importlib.util.spec_from_file_location("training_entry", os.path.join(ROOT, "scripts", "training.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    logs = {}
    for mode in ("auto", "0"):
        monkeypatch.setenv("BTVAE_STEP_GRAPH", mode)
        cfg = tr.load_config(os.path.join(ROOT, "config_files", "config0.json"))
        cfg["train"].update({"epochs": 1, "device": "cuda:0", "save_epoch_freq": None, "verbose_freq": 250})
        cfg["save_dir"] = str(tmp_path / mode)
        os.makedirs(cfg["save_dir"])
        seen = logs.setdefault(mode, [])
        monkeypatch.setattr(VM.VAE, "train_verbose", lambda self, itx, loss, pbar=None, seen=seen: seen.append((itx, float(loss["elbo"]), float(loss["kl"]))))
        torch.manual_seed(2112)
        model = tr.main(cfg, n_train=1000 * cfg["train"]["batch_size"], n_val=8192)
        assert (len(model.__dict__.get("_step_graphs", {})) > 0) == (mode == "auto")
    a, b = np.array(logs["auto"]), np.array(logs["0"])
    assert a.shape == b.shape and a.shape[0] >= 4 and (a[:, 0] == b[:, 0]).all()
    assert np.abs(a[:, 1:] - b[:, 1:]).max() <= 5e-3 * np.abs(b[:, 1:]).max(), (a, b)
    assert a[-1, 1] > a[0, 1] and a[-1, 2] > 5.0, a


def test_training_entry_point_on_arrays(dev, tmp_path):
    """``scripts/training.py --train_npz/--val_npz``: real arrays instead of the synthetic fallback, including a
    day-before profile tag (identity) next to the calendar tags, so the condition width follows the data (2+2+24)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("training_entry", os.path.join(ROOT, "scripts", "training.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    rng = np.random.default_rng(5)

    def split(n, path):
        x = np.clip(rng.normal(1.0, 1.0, size=(n, 24)), -3, 5).astype(np.float32)
        np.savez(path, X=x, cond_months=rng.integers(1, 13, size=n), cond_weekdays=rng.integers(0, 7, size=(n, 1)),
                 cond_day_befores=np.roll(x, 1, axis=0))
    split(1024, tmp_path / "train.npz")
    split(300, tmp_path / "val.npz")
    cfg = tr.load_config(os.path.join(ROOT, "config_files", "config0.json"))
    cfg["data"]["condition_tag_list"] = ["months", "weekdays", "day_befores"]
    cfg["model"]["latent_dim"] = 12
    for k in ("posterior", "likelihood"):
        cfg["model"]["distribution_dict"][k]["num_neurons"] = 64
    cfg["train"].update({"epochs": 1, "batch_size": 256, "validation_freq": 2, "verbose_freq": 100, "num_mc_samples": 2,
                         "validation_mc_samples": 2, "device": "cuda:0", "save_epoch_freq": 1})
    cfg["save_dir"] = str(tmp_path)
    model = tr.main(cfg, train_npz=str(tmp_path / "train.npz"), val_npz=str(tmp_path / "val.npz"))
    assert model.input_dim == 24 and model.condition_dim == 28
    sd = torch.load(os.path.join(model.log_dir, "trained_model.pt"), weights_only=True)
    assert sd["encoder.parameterizer.block_dict.input.input_layer.0.weight"].shape == (64, 52)
    assert int(sd["encoder.parameterizer.block_dict.mu.input_layer.2.num_batches_tracked"]) == 4
    with pytest.raises(ValueError, match="missing condition arrays"):
        cfg["data"]["condition_tag_list"] = ["months", "weekdays", "users"]
        tr.main(cfg, train_npz=str(tmp_path / "train.npz"))


def test_device_condition_feed_dirichlet(dev):
    """Sampled (Dirichlet "sample" style) conditions drawn on the device.  The deterministic tags' columns equal the
    host conditioner bit for bit; the sampled block is a Dirichlet(gamma) draw -- rows sum to 1, means and variances
    match the closed forms, draws are reproducible under torch.manual_seed and differ call to call.  Parity with the
    reference's numpy stream is unpinned by construction (different generator); the check is distributional."""
    from behavio_temporal_vae_amd.conditioning_lib import Conditioner
    from behavio_temporal_vae_amd.datasets import ConditionedDataset, DeviceConditionEncoder
    from behavio_temporal_vae_amd.loaders import DeviceBatchLoader
    rng = np.random.default_rng(0)
    n, T, S = 64, 37, 4000
    months = rng.integers(1, 13, size=(n, 1)); weekdays = rng.integers(0, 7, size=(n, 1))
    day_before = rng.normal(size=(n, 3))
    gammas = np.where(rng.random((n, T)) < 0.5, 0.01, 0.01 + rng.gamma(2.0, 3.0, size=(n, T)))   # sparse LDA-like rows
    cset = {"months": months, "weekdays": weekdays, "users": gammas, "day_befores": day_before}
    cond = Conditioner(tags=["months", "weekdays", "users", "day_befores"],
                       supports=[list(range(1, 13)), list(range(7)), [0.01, 50.0], [-3, 3]],
                       types=["circ", "circ", "dir", "identity"], condition_set=cset)
    assert cond.is_random() and cond.cond_dim == 4 + T + 3
    enc = DeviceConditionEncoder(cond, cset, dev)
    idx = torch.tensor([5, 0, 63, 17], device=dev)
    torch.manual_seed(7)
    c = enc.encode(idx, S)
    assert c.shape == (4, S, cond.cond_dim)
    host = cond.transform({k: v[idx.cpu().numpy()] for k, v in cset.items()}, num_samples=2)   # [2, 4, C]
    det = np.r_[0:4, 4 + T:4 + T + 3]
    got = c.cpu().numpy()
    assert np.array_equal(got[:, 0][:, det], host[0][:, det].astype(np.float32))
    assert np.array_equal(got[:, -1][:, det], host[0][:, det].astype(np.float32))
    d = got[:, :, 4:4 + T].astype(np.float64)
    assert np.all(d >= 0) and np.allclose(d.sum(-1), 1.0, atol=1e-5)
    g = gammas[idx.cpu().numpy()]
    a0 = g.sum(1, keepdims=True)
    mean, var = g / a0, g * (a0 - g) / (a0 ** 2 * (a0 + 1))
    assert np.abs(d.mean(1) - mean).max() < 6 * np.sqrt(var.max() / S) + 1e-4
    big = var > 1e-5
    assert np.abs(d.var(1)[big] / var[big] - 1).max() < 0.2
    enc2 = DeviceConditionEncoder(cond, cset, dev)
    torch.manual_seed(7)
    assert torch.equal(enc2.encode(idx, S), c)                       # same seed, same call number
    assert not torch.equal(enc2.encode(idx, S)[..., 4:4 + T], c[..., 4:4 + T])
    # the loader hands [B, S, C] batches to a model whose forward takes 3-D conditions
    ds = ConditionedDataset(rng.normal(size=(n, 6)).astype(np.float32), cset, cond, num_samples=3)
    x, cb = next(iter(DeviceBatchLoader(ds, 16, shuffle=True, drop_last=True, device=dev)))
    assert x.shape == (16, 6) and cb.shape == (16, 3, cond.cond_dim) and cb.is_cuda


def test_streamed_test_metrics_vs_reference(dev, tmp_path):
    """Per-profile IW log-likelihood, KL and l1/l2 sample errors as the reference's collection script derives them
    from its host buffers (scripts/collect_testing_results.py:84-113), here streamed on the device with the same
    draws injected; then the two result files with the reference's columns / keys."""
    import pickle
    import pandas as pd
    from behavio_temporal_vae_amd import results_lib, testing_lib
    g = load_golden("test_metrics_tiny")
    cond, _ = make_cond(g)
    m = build(g, dev, cond)
    x, c = T(g["x"]).to(dev), T(g["c"]).to(dev)
    res = testing_lib.test_metrics_device(m, x, c, int(g["K"]), mc_sample_batch_size=4, eps=T(g["eps"]).to(dev), eps_x=T(g["eps_x"]).to(dev))
    for k in ("loglikelihood", "kl", "l_1", "l_2"):
        assert rel_err(res[k].cpu().numpy(), g[k]) < 1e-4, k
    metrics = {k: v.cpu().numpy() for k, v in res.items()}
    users, months = np.arange(int(g["N"])) // 3, g["months"][:, 0]
    frame = results_lib.test_results_frame(metrics, users, months, "test")
    agg = results_lib.test_results_aggregate(metrics)
    results_lib.save_test_results(str(tmp_path), {"test": frame}, {"test": agg})
    back = pd.read_csv(tmp_path / "test_results_raw.csv")
    assert list(back.columns) == ["l_1", "l_2", "loglikelihood", "user_id", "month", "set_type"]
    assert np.allclose(back["loglikelihood"].values, g["loglikelihood"], rtol=1e-4) and set(back["set_type"]) == {"test"}
    with open(tmp_path / "test_results_aggregate.pkl", "rb") as f:
        a = pickle.load(f)
    assert set(a["test"]) == {"loglikelihood", "error_l_1", "error_l_2"}
    assert abs(a["test"]["error_l_2"] - float(g["l_2"].mean())) <= 1e-4 * abs(float(g["l_2"].mean()))


def test_collection_scripts_end_to_end(dev, tmp_path):
    """Both evaluation CLIs on a saved run folder (trained_model.pt + config.json + conditioner.pkl) and an .npz test
    split: the files, columns and keys the reference's scripts write (scripts/collect_testing_results.py:117-137,
    scripts/collect_imputation_results.py:112-148), and the imputation errors recomputed from the saved samples."""
    import importlib.util
    import pickle
    import types
    import pandas as pd
    g = load_golden("test_metrics_tiny")
    cond, cset = make_cond(g)
    m = build(g, dev, cond)
    run = tmp_path / "runs" / "run0"
    run.mkdir(parents=True)
    m.save(str(run), device=str(dev))
    cond.save(str(run))
    with open(run / "config.json", "w") as f:
        json.dump({"model": json.loads(str(g["model_cfg"]))}, f)
    (tmp_path / "runs" / "not_a_run").mkdir()
    N, D = int(g["N"]), int(g["D"])
    npz = tmp_path / "test.npz"
    np.savez(npz, X_test=g["x"], user_ids=np.arange(N) // 3, months=g["months"][:, 0], cond_months=g["months"], cond_weekdays=g["weekdays"])

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(__file__), "..", "scripts", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    common = dict(config_dir=str(tmp_path / "runs"), test_npz=str(npz), overwrite=True, device=str(dev))
    done = load("collect_testing_results").main(types.SimpleNamespace(num_rec_samples=6, batch_size=5, batch_size_mc=4, **common))
    assert set(done) == {"run0"}
    raw = pd.read_csv(run / "test_results_raw.csv")
    assert list(raw.columns) == ["l_1", "l_2", "loglikelihood", "user_id", "month", "set_type"] and len(raw) == N

    W = D // 2
    args = types.SimpleNamespace(num_imputation_samples=4, batch_size=5, unobserved_window_length=W, num_pseudo_gibbs_steps=6,
                                 num_metropolis_within_hasting_steps=0, **common)
    done = load("collect_imputation_results").main(args)
    assert set(done) == {"run0"} and set(done["run0"]) == {"loglikelihood", "error_l_1", "error_l_2"}
    sub = run / f"forecasting_results_{W}"
    raw = pd.read_csv(sub / "test_results_raw.csv")
    assert list(raw.columns) == ["l_1", "l_2", "loglikelihood", "user_id", "month", "set_type"] and len(raw) == N
    assert np.isfinite(raw[["l_1", "l_2", "loglikelihood"]].values).all()
    with open(sub / "test_results_aggregate.pkl", "rb") as f:
        agg = pickle.load(f)
    assert abs(agg["test"]["error_l_1"] - raw["l_1"].mean()) < 1e-5 * abs(raw["l_1"].mean())
    with open(sub / "test_results_samples.pkl", "rb") as f:
        smp = pickle.load(f)
    assert smp["test"].shape == (N, W)
    # second call without overwrite skips the folder
    args.overwrite = False
    assert load("collect_imputation_results").main(args) == {}


def test_eval_chain_in_pair_format_equals_fp32_chain(dev):
    """The eval-mode chains keep their hidden activations in the pair format from a row count on (``engine.EVAL_PAIRS_MIN_ROWS``):
    with the threshold at zero the importance-weighted scoring of a gipuzkoa-shaped model must reproduce the fp32-activation
    chain (the pieces the products multiply are the same ones: 1e-5 on the per-draw log-likelihoods, which are O(100))."""
    from behavio_temporal_vae_amd import engine as E, testing_lib
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = {"latent_dim": 24, "distribution_dict": {
        "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": 200,
                      "num_hidden_layers": 2, "dropout": True, "dropout_rate": 0.2, "batch_normalization": True},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "sigma_lim": 1e-6, "marginal_std_lim": 0.05, "mu_lower_lim": -3,
                       "num_neurons": 200, "num_hidden_layers": 3, "dropout": True, "dropout_rate": 0.2, "batch_normalization": True}}}

    class C4:
        cond_dim = 4
    torch.manual_seed(5)
    m = CVAE(input_dim=24, conditioner=C4(), **cfg).to(dev).eval()
    assert m.gemm_precision == "fp32_split2"
    N, K = 300, 40
    gen = torch.Generator().manual_seed(6)
    x, c = torch.randn(N, 24, generator=gen).to(dev), torch.randn(N, 4, generator=gen).to(dev)
    eps = torch.randn(K, N, 24, generator=gen).to(dev)
    keep = E.EVAL_PAIRS_MIN_ROWS
    seen = []
    orig = E.Pairs.__init__

    def spy(self, data, sc):
        seen.append(tuple(data.shape))
        orig(self, data, sc)
    try:
        E.EVAL_PAIRS_MIN_ROWS = 1 << 30
        a, ra, _ = testing_lib.iw_loglikelihood_device(m, x, c, K, mc_sample_batch_size=20, eps=eps, return_rlls=True)
        E.EVAL_PAIRS_MIN_ROWS = 0
        E.Pairs.__init__ = spy
        b, rb, _ = testing_lib.iw_loglikelihood_device(m, x, c, K, mc_sample_batch_size=20, eps=eps, return_rlls=True)
    finally:
        E.EVAL_PAIRS_MIN_ROWS = keep
        E.Pairs.__init__ = orig
    assert (N * 20, 200) in seen, seen              # the decoder's trunk handed pair-format activations on
    assert torch.allclose(ra, rb, rtol=1e-6, atol=1e-5 * float(ra.abs().max()))
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-5 * float(a.abs().max()))


@pytest.mark.parametrize("strict", [True, False])
def test_eval_pair_chain_stays_fp32_behind_a_flagged_input(dev, strict):
    """A pair-format activation has one scale per 128 x 128 block and no three-piece form.  A decoder input with one row 2^24
    times the others (here: one profile's conditions) gives hidden activations with such a row, next to which the other rows of
    its block would keep nothing of their values.  The chain starts only while the range flag of its first product is down: with the test made
    before every product (strict) no pair-format activation is ever written and the scores of the other profiles meet the
    fp32-instruction chain at 1e-4; without waiting, the first call raises the flag and the next one stays fp32."""
    from behavio_temporal_vae_amd import engine as E, testing_lib
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = {"latent_dim": 24, "distribution_dict": {
        "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": 200,
                      "num_hidden_layers": 2, "dropout": False, "batch_normalization": True},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "sigma_lim": 1e-6, "marginal_std_lim": 0.05, "mu_lower_lim": -3,
                       "num_neurons": 200, "num_hidden_layers": 3, "dropout": False, "batch_normalization": True}}}

    class C4:
        cond_dim = 4
    torch.manual_seed(5)
    m = CVAE(input_dim=24, conditioner=C4(), **cfg).to(dev).eval()
    assert m.gemm_precision == "fp32_split2"
    N, K = 300, 40
    gen = torch.Generator().manual_seed(16)
    x, c = torch.randn(N, 24, generator=gen).to(dev), torch.randn(N, 4, generator=gen).to(dev)
    c[7] *= 2.0 ** 24
    eps = torch.randn(K, N, 24, generator=gen).to(dev)
    others = torch.ones(N, dtype=torch.bool, device=dev)
    others[7] = False
    keep_rows, was = E.EVAL_PAIRS_MIN_ROWS, E.GUARD.strict
    seen = []
    orig = E.Pairs.__init__

    def spy(self, data, sc):
        seen.append(tuple(data.shape))
        orig(self, data, sc)
    try:
        m.set_gemm_precision("fp32")
        ref, rref, _ = testing_lib.iw_loglikelihood_device(m, x, c, K, mc_sample_batch_size=20, eps=eps, return_rlls=True)
        m.set_gemm_precision("fp32_split2")
        E.GUARD.reset()
        E.GUARD.strict = strict
        E.EVAL_PAIRS_MIN_ROWS = 0
        E.Pairs.__init__ = spy
        got, rgot, _ = testing_lib.iw_loglikelihood_device(m, x, c, K, mc_sample_batch_size=20, eps=eps, return_rlls=True)
        torch.cuda.synchronize()
        assert E.GUARD.tripped_sites() > 0
        if not strict:
            assert seen, "the first chunks ran before the host saw a flag"
            del seen[:]
            got, rgot, _ = testing_lib.iw_loglikelihood_device(m, x, c, K, mc_sample_batch_size=20, eps=eps, return_rlls=True)
        assert not seen, seen                        # no pair-format activation behind the flagged product
        assert E.GUARD.fallbacks > 0
        E.GUARD.check()
    finally:
        E.EVAL_PAIRS_MIN_ROWS, E.GUARD.strict = keep_rows, was
        E.Pairs.__init__ = orig
        E.GUARD.reset()
        m.set_gemm_precision("fp32_split2")
    assert torch.allclose(rgot[:, others], rref[:, others], rtol=1e-4, atol=1e-4 * float(rref[:, others].abs().max())), \
        float((rgot[:, others] - rref[:, others]).abs().max())
    assert torch.allclose(got[others], ref[others], rtol=1e-4, atol=1e-4 * float(ref[others].abs().max()))
