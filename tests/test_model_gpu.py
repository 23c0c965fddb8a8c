"""GPU: the model-level HIP path (forward, ELBO, backward, clip+Adam) against the golden vectors produced
by the reference and against the CPU oracle on the same seeded inputs.  Tolerance for ELBO/KL/RLL terms is
1e-4 relative (fp32), as the task states; gradients are compared at 5e-4 of each tensor's max magnitude."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_spec, golden_state, load_golden, rel_err

pytestmark = pytest.mark.gpu


class Cdim:
    def __init__(self, c):
        self.cond_dim = c


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def build(g, dev, seed=0):
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = json.loads(str(g["model_cfg"]))
    torch.manual_seed(seed)
    m = CVAE(input_dim=int(g["D"]), conditioner=Cdim(int(g["C"]) if "C" in g else 4), **cfg)
    sd = golden_state(g)
    if sd:
        m.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.parametrize("name", ["tiny_dict_2d", "tiny_dict_3d", "tiny_normal_2d"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_forward_loss_vs_reference(dev, name, mode):
    g = load_golden(name)
    m = build(g, dev)
    m.train() if mode == "train" else m.eval()
    m.prior_params = {k: v.to(dev) for k, v in m.prior_params.items()}
    x, c, eps = T(g["x"]).to(dev), T(g["c"]).to(dev), T(g[f"{mode}_eps"]).to(dev)
    S = eps.shape[0]
    fw = m._run_forward(x, c, S, eps=eps)
    xd, zd = m._pack_outputs(fw)
    assert np.allclose(zd["params"]["mu"].cpu().numpy(), g[f"{mode}_post_mu"], rtol=1e-4, atol=2e-5)
    assert np.allclose(zd["params"]["sigma"].cpu().numpy(), g[f"{mode}_post_sigma"], rtol=1e-4, atol=2e-5)
    assert np.allclose(zd["samples"].cpu().numpy(), g[f"{mode}_z"], rtol=1e-4, atol=2e-5)
    assert np.allclose(xd["params"]["mu"].cpu().numpy(), g[f"{mode}_lik_mu"], rtol=1e-4, atol=5e-5)
    assert np.allclose(xd["params"]["sigma"].cpu().numpy(), g[f"{mode}_lik_sigma"], rtol=1e-4, atol=5e-5)
    rows = m.reconstruction_loglikelihood(x, xd["params"])
    assert rel_err(rows.cpu().numpy(), g[f"{mode}_rll_rows"]) < 1e-4
    loss = m.loss(x, xd["params"], zd["params"], beta=0.7, prior_params=m.prior_params)
    for k in ("loss", "elbo", "rll", "kl"):
        assert abs(loss[k].item() - float(g[f"{mode}_{k}"])) <= 1e-4 * abs(float(g[f"{mode}_{k}"])) + 1e-5, k
    fused = m._loss_from_forward(fw, x, 0.7).tolist()
    for v, k in zip(fused, ("loss", "elbo", "rll", "kl")):
        assert abs(v - float(g[f"{mode}_{k}"])) <= 1e-4 * abs(float(g[f"{mode}_{k}"])) + 1e-5, k
    if mode == "train":
        sd = m.state_dict()
        for k, v in g.items():
            if k.startswith("after_fwd_sd::"):
                got = sd[k[len("after_fwd_sd::"):]].cpu().numpy()
                assert np.allclose(got, v, rtol=1e-5, atol=1e-6), k
    if mode == "eval" and name.startswith("tiny_dict"):
        cov = m.decoder.create_covariance_matrix(xd["params"])
        assert np.allclose(cov.cpu().numpy(), g["eval_cov"], rtol=1e-4, atol=1e-6)
        # rsample parity through the product path (decoder.rsample -> btvae_matmul_nt_add) with the recorded noise
        e2 = T(g["eval_dict_rsample_eps"]).to(dev)
        with _FixedRandn(e2):
            got = m.decoder.rsample(xd["params"], num_samples=e2.shape[0])
        assert got.shape == tuple(g["eval_dict_rsample"].shape)
        assert np.allclose(got.cpu().numpy(), g["eval_dict_rsample"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("name", ["tiny_dict_2d", "tiny_normal_2d"])
def test_gradients_vs_reference(dev, name):
    g = load_golden(name)
    m = build(g, dev)
    m.train()
    x, c, eps = T(g["x"]).to(dev), T(g["c"]).to(dev), T(g["train_eps"]).to(dev)
    fw = m._run_forward(x, c, eps.shape[0], eps=eps, tape=True)
    out = m._loss_and_backward(fw, x, 0.7).tolist()
    assert abs(out[0] - float(g["train_loss"])) <= 1e-4 * abs(float(g["train_loss"]))
    for k, p in m.named_parameters():
        ref = g["grad::" + k]
        scale = np.abs(ref).max() + 1e-8
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 5e-4 * scale + 1e-6, (k, err, scale)


@pytest.mark.parametrize("name", ["tiny_dict_2d", "tiny_normal_2d"])
def test_train_steps_vs_reference(dev, name):
    g = load_golden(name)
    m = build(g, dev)
    m.train()
    m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
    for t in range(g["steps_x"].shape[0]):
        loss = m._train_step(T(g["steps_x"][t]).to(dev), T(g["steps_c"][t]).to(dev), opt, eps=T(g["steps_eps"][t]).to(dev))
        got = [loss[k].item() for k in ("loss", "elbo", "rll", "kl")]
        assert rel_err(got, g["steps_losses"][t]) < 1e-4, (t, got, g["steps_losses"][t])
    sd = m.state_dict()
    for k, v in sd.items():
        ref = g["final::" + k]
        if v.is_floating_point():
            assert np.abs(v.cpu().numpy() - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-6), k
        else:
            assert int(v) == int(ref), k


def _keep(g, prefix, dev):
    out, i = [], 0
    while f"{prefix}_keep_{i}" in g:
        out.append(T(g[f"{prefix}_keep_{i}"]).to(dev).contiguous())
        i += 1
    return out


@pytest.mark.parametrize("name", ["tiny_dropout_2d", "tiny_nobn_2d"])
def test_dropout_training_vs_reference(dev, name):
    """Train-mode dropout (src/submodels.py:23-25,51) with the reference's recorded keep flags injected, for units
    with and without BatchNorm: loss terms and every gradient of one step, then four optimiser steps and the
    final state."""
    from behavio_temporal_vae_amd import engine as E
    g = load_golden(name)
    m = build(g, dev)
    m.train()
    x, c, eps = T(g["x"]).to(dev), T(g["c"]).to(dev), T(g["train_eps"]).to(dev)
    try:
        E.DROPOUT.injected = _keep(g, "train", dev)
        fw = m._run_forward(x, c, eps.shape[0], eps=eps, tape=True)
        assert not E.DROPOUT.injected
        out = m._loss_and_backward(fw, x, 0.7).tolist()
        for v, k in zip(out, ("loss", "elbo", "rll", "kl")):
            assert abs(v - float(g[f"train_{k}"])) <= 1e-4 * abs(float(g[f"train_{k}"])) + 1e-5, k
        for k, p in m.named_parameters():
            ref = g["grad::" + k]
            scale = np.abs(ref).max() + 1e-8
            err = np.abs(p.grad.cpu().numpy() - ref).max()
            assert err <= 5e-4 * scale + 1e-6, (k, err, scale)
        m = build(g, dev)
        m.train()
        m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
        opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
        for t in range(g["steps_x"].shape[0]):
            E.DROPOUT.injected = _keep(g, f"step{t}", dev)
            loss = m._train_step(T(g["steps_x"][t]).to(dev), T(g["steps_c"][t]).to(dev), opt, eps=T(g["steps_eps"][t]).to(dev))
            got = [loss[k].item() for k in ("loss", "elbo", "rll", "kl")]
            assert rel_err(got, g["steps_losses"][t]) < 1e-4, (t, got, g["steps_losses"][t])
    finally:
        E.DROPOUT.injected = None
    sd = m.state_dict()
    for k, v in sd.items():
        ref = g["final::" + k]
        if v.is_floating_point():
            assert np.abs(v.cpu().numpy() - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-6), k


def test_dropout_philox_flags(dev):
    """Device-drawn keep flags: Bernoulli(1-p) frequency, reproducible under torch.manual_seed, different
    between consecutive draws; a dropout model trains and its eval forward is deterministic."""
    from behavio_temporal_vae_amd import engine as E
    torch.manual_seed(5)
    a = E.DROPOUT.keep(4096, 1000, 0.3, dev)
    b = E.DROPOUT.keep(4096, 1000, 0.3, dev)
    torch.manual_seed(5)
    a2 = E.DROPOUT.keep(4096, 1000, 0.3, dev)
    assert torch.equal(a, a2) and not torch.equal(a, b)
    assert set(a.unique().tolist()) == {0, 1}
    assert abs(a.float().mean().item() - 0.7) < 2e-3
    assert abs(a.float().mean(0).min().item() - 0.7) < 0.05 and abs(a.float().mean(1).min().item() - 0.7) < 0.08
    g = load_golden("tiny_dropout_2d")
    m = build(g, dev)
    m.train()
    m.train_kwargs = {"num_mc_samples": 3, "beta": 1.0, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
    x, c = T(g["x"]).to(dev), T(g["c"]).to(dev)
    l0 = m._train_step(x, c, opt)["loss"].item()
    for _ in range(30):
        l1 = m._train_step(x, c, opt)["loss"].item()
    assert np.isfinite(l1) and l1 < l0


def test_config0_shape_train_steps(dev):
    """config0 architecture (H=1000, D=48, L=48, V=100), weights from the seeded constructor: pins seed-exact
    initialisation, the per-step ELBO terms (1e-4) and parameter slices after 3 optimiser steps."""
    g = load_golden("config0_smallbatch")
    m = build(g, dev, seed=int(g["seed"]))
    sd0 = m.state_dict()
    for k, v in g.items():
        if k.startswith("init::"):
            assert np.array_equal(sd0[k[6:]].cpu().numpy().reshape(-1)[:64], v), k
    # fp64 oracle trajectory from the same initial weights: the arbiter for step 3.  Adam's first steps are
    # sign-like, so fp32 round-off is amplified: the reference's own fp32 CPU run is 2.4e-3 away from the exact
    # (fp64) trajectory at step 3 while agreeing to <3e-5 at steps 1-2.  Steps 1-2 are therefore held to 1e-4
    # against both the reference's numbers and the fp64 trajectory; step 3 to 3e-3 against the fp64 trajectory, the
    # band the reference's own fp32 run falls in (three evaluation orders of this kernel set gave 5e-5, 1.7e-4, 1.9e-3).
    from oracle import refimpl as R
    cfg = json.loads(str(g["model_cfg"]))
    spec = R.model_spec(48, 4, cfg)
    sd64 = R.cast_state({k: v.detach().cpu().clone() for k, v in sd0.items()}, torch.float64)
    st = R.AdamState()
    m.train()
    m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": float(g["beta"]), "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt = m.get_optimizer(lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    for t in range(g["steps_x"].shape[0]):
        x, c, eps = T(g["steps_x"][t]), T(g["steps_c"][t]), T(g["steps_eps"][t])
        ref64, _, _ = R.train_step(sd64, spec, st, x.double(), c.double(), eps.double(), lr=float(g["lr"]),
                                   weight_decay=float(g["weight_decay"]), beta=float(g["beta"]), max_norm=1.0)
        loss = m._train_step(x.to(dev), c.to(dev), opt, eps=eps.to(dev))
        got = [loss[k].item() for k in ("loss", "elbo", "rll", "kl")]
        assert rel_err(got, [ref64[k] for k in ("loss", "elbo", "rll", "kl")]) < (1e-4 if t < 2 else 3e-3), (t, got, ref64)
        if t < 2:
            assert rel_err(got, g["steps_losses"][t]) < 1e-4, (t, got, g["steps_losses"][t])
    sd1 = m.state_dict()
    for k in ("encoder.parameterizer.block_dict.input.input_layer.0.weight", "decoder.parameterizer.block_dict.input.middle_layers.1.0.weight",
              "decoder.SigmaMapper", "decoder.parameterizer.block_dict.mu.input_layer.2.running_var"):
        d = np.abs(sd1[k].cpu().numpy().astype(np.float64) - sd64[k].numpy())
        # a parameter moves at most lr per step; sign flips of near-zero gradients may move single entries 2*lr apart
        assert np.median(d) < 3e-4 and d.max() < 7e-3, (k, np.median(d), d.max())


@pytest.mark.parametrize("hidden", [0, 1, 2])
def test_train_core_against_oracle_random_shapes(dev, hidden):
    """Odd sizes (rows not multiples of the 128 tile, widths not multiples of 4) against the fp32 CPU oracle, for
    0, 1 and 2 residual middle layers: with none, every BatchNorm output but the heads' feeds a plain Linear and
    stays unmaterialised."""
    from oracle import refimpl as R
    from behavio_temporal_vae_amd.vae_models import CVAE
    cfg = {"latent_dim": 7, "distribution_dict": {
        "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": 70,
                      "num_hidden_layers": hidden, "dropout": False, "batch_normalization": True},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 13, "sigma_lim": 1e-4, "marginal_std_lim": 0.1, "mu_lower_lim": -3,
                       "num_neurons": 70, "num_hidden_layers": hidden, "dropout": False, "batch_normalization": True}}}
    D, C, B, S = 10, 3, 150, 3
    torch.manual_seed(5)
    m = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    spec = R.model_spec(D, C, cfg)
    m = m.to(dev).train()
    m.train_kwargs = {"num_mc_samples": S, "beta": 1.0, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
    st = R.AdamState()
    gen = torch.Generator().manual_seed(9)
    for t in range(3):
        x = torch.randn(B, D, generator=gen).clamp(-3, 5)
        c = torch.randn(B, C, generator=gen)
        eps = torch.randn(S, B, 7, generator=gen)
        ref, _, _ = R.train_step(sd, spec, st, x, c, eps, lr=1e-3, weight_decay=1e-5, beta=1.0, max_norm=1.0)
        loss = m._train_step(x.to(dev), c.to(dev), opt, eps=eps.to(dev))
        got = [loss[k].item() for k in ("loss", "elbo", "rll", "kl")]
        assert rel_err(got, [ref[k] for k in ("loss", "elbo", "rll", "kl")]) < 1e-4, (t, got, ref)


def test_unmaterialised_batchnorm_outputs_change_nothing(dev):
    """Folding a BatchNorm into the Linear that reads it is a re-association only: the loss terms of two optimiser
    steps and every gradient of the second agree with the materialising path to fp32 round-off (parameters are not
    compared: Adam's first update is sign-like, so a weight whose gradient is round-off-sized may move either way)."""
    from behavio_temporal_vae_amd import engine as E
    from behavio_temporal_vae_amd.vae_models import CVAE
    g = load_golden("tiny_dict_2d")
    res = {}
    for fold in (True, False):
        E.FOLD_BN_INTO_CONSUMER = fold
        try:
            m = build(g, dev).train()
            m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
            opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
            losses = []
            for t in range(2):
                l = m._train_step(T(g["steps_x"][t]).to(dev), T(g["steps_c"][t]).to(dev), opt, eps=T(g["steps_eps"][t]).to(dev))
                losses.append([l[k].item() for k in ("loss", "elbo", "rll", "kl")])
            grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
            res[fold] = (np.array(losses), grads)
        finally:
            E.FOLD_BN_INTO_CONSUMER = True
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    for n, gr in res[False][1].items():
        assert (res[True][1][n] - gr).abs().max() <= 1e-4 * (gr.abs().max() + 1e-8), n


def test_wide_conditioning_3d_fp32_and_bf16(dev):
    """Sweep-style shape (scripts/config_sweep.py:76-81): months + weekdays + day-before profile + T=500 user topics
    -> C = 528, D = L = 24, sampled (3-D) conditions [B, S, C].  fp32 MFMA path vs the fp32 oracle at 1e-4; the
    bf16-operand path at 1e-2 on every ELBO term (forward, eval and train mode) and on a three-step loss trajectory."""
    from oracle import refimpl as R
    from behavio_temporal_vae_amd.vae_models import CVAE
    H = 200
    cfg = {"latent_dim": 24, "distribution_dict": {
        "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": H,
                      "num_hidden_layers": 2, "dropout": False, "batch_normalization": True},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "sigma_lim": 1e-4, "marginal_std_lim": 0.1, "mu_lower_lim": -3,
                       "num_neurons": H, "num_hidden_layers": 2, "dropout": False, "batch_normalization": True}}}
    D, C, B, S = 24, 528, 192, 4
    torch.manual_seed(11)
    m = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    spec = R.model_spec(D, C, cfg)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, D, generator=gen).clamp(-3, 5)
    topics = torch.distributions.Dirichlet(torch.full((500,), 0.05)).sample((B, S))
    c = torch.cat([torch.randn(B, S, 28, generator=gen), topics], -1).float()
    eps = torch.randn(S, B, 24, generator=gen)
    m = m.to(dev)
    for train in (True, False):
        lik, post, _ = R.cvae_forward(dict(sd), spec, x, c, eps, train, {})
        ref = {k: float(v) for k, v in R.elbo_terms(sd, spec, x, lik, post, beta=1.0).items()}
        for prec, tol in (("fp32", 1e-4), ("bf16", 1e-2)):
            m.load_state_dict(sd)
            m.train() if train else m.eval()
            m.set_gemm_precision(prec)
            fw = m._run_forward(x.to(dev), c.to(dev), S, eps=eps.to(dev))
            got = m._loss_from_forward(fw, x.to(dev), 1.0).tolist()
            for v, k in zip(got, ("loss", "elbo", "rll", "kl")):
                assert abs(v - ref[k]) <= tol * abs(ref[k]) + tol * 1e-1, (train, prec, k, v, ref[k])
    # training (2-D conditions, as train_core feeds them) in bf16 follows the fp32 oracle's loss trajectory to 1e-2
    c2 = c[:, 0].contiguous()
    m.load_state_dict(sd)
    m.train()
    m.set_gemm_precision("bf16")
    m.train_kwargs = {"num_mc_samples": S, "beta": 1.0, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
    st = R.AdamState()
    sdo = {k: v.clone() for k, v in sd.items()}
    for t in range(3):
        ref, _, _ = R.train_step(sdo, spec, st, x, c2, eps, lr=1e-3, weight_decay=1e-5, beta=1.0, max_norm=1.0)
        loss = m._train_step(x.to(dev), c2.to(dev), opt, eps=eps.to(dev))
        got = [loss[k].item() for k in ("loss", "elbo", "rll", "kl")]
        assert rel_err(got, [ref[k] for k in ("loss", "elbo", "rll", "kl")]) < 1e-2, (t, got, ref)
    m.set_gemm_precision("fp32")
    from behavio_temporal_vae_amd import _native as nv
    nv.set_gemm_precision("fp32")


@pytest.mark.timeout(600)
def test_wide_conditioning_at_benchmark_size(dev):
    """The wide-conditioning leg at the size the benchmark runs it -- C = 528, D = L = 24, H = 1000, B = 4096, S = 16
    (65,536 decoder rows, first-layer reductions of 552) -- train-mode forward and ELBO terms: bf16 operands within 1e-2
    of the fp32 oracle, the default fp32-accurate mode within 1e-4."""
    from oracle import refimpl as R
    from behavio_temporal_vae_amd.vae_models import CVAE
    import bench
    D, C, B, S = 24, 528, 4096, 16
    torch.manual_seed(12)
    m = CVAE(input_dim=D, conditioner=Cdim(C), **bench.WIDE_MODEL)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    cfg = {k: bench.WIDE_MODEL[k] for k in ("latent_dim", "distribution_dict")}
    spec = R.model_spec(D, C, cfg)
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(B, D, generator=gen).add_(1.0).clamp_(-3, 5)
    topics = torch.distributions.Dirichlet(torch.full((500,), 0.05)).sample((B,))
    c = torch.cat([torch.randn(B, 28, generator=gen), topics], -1).float()
    eps = torch.randn(S, B, 24, generator=gen)
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    lik, post, _ = R.cvae_forward(dict(sd), spec, x, c, eps, True, {})
    ref = {k: float(v) for k, v in R.elbo_terms(sd, spec, x, lik, post, beta=1.0).items()}
    m = m.to(dev).train()
    try:
        for prec, tol in (("bf16", 1e-2), ("fp32_split2", 1e-4)):
            m.load_state_dict(sd)
            m.set_gemm_precision(prec)
            fw = m._run_forward(x.to(dev), c.to(dev), S, eps=eps.to(dev))
            got = m._loss_from_forward(fw, x.to(dev), 1.0).tolist()
            for v, k in zip(got, ("loss", "elbo", "rll", "kl")):
                assert abs(v - ref[k]) <= tol * abs(ref[k]) + tol * 1e-1, (prec, k, v, ref[k])
    finally:
        m.set_gemm_precision("fp32_split2")


@pytest.mark.parametrize("bn,dropout,prec", [(True, False, "fp32_split2"), (True, True, "fp32_split"), (False, True, "fp32_split2"), (False, False, "fp32")])
def test_padded_activation_rows_change_no_bits(dev, monkeypatch, bn, dropout, prec):
    """Hidden activations and gradients in rows padded to a multiple of 32 floats (``engine._rows_on_lines``; width 72 ->
    pitch 96 here) against dense ones: the same kernels on the same values, so three training steps (losses, weights,
    BatchNorm buffers) and an eval-mode forward agree to the last bit -- with and without BatchNorm, with dropout (whose
    keep flags stay dense) and in every fp32-accurate GEMM mode."""
    from behavio_temporal_vae_amd import engine as E
    from behavio_temporal_vae_amd.vae_models import CVAE
    H, D, C, B, S = 72, 24, 6, 200, 3
    cfg = {"latent_dim": 10, "gemm_precision": prec, "distribution_dict": {
        "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": H,
                      "num_hidden_layers": 2, "dropout": dropout, "dropout_rate": 0.2, "batch_normalization": bn},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 40, "sigma_lim": 1e-4, "marginal_std_lim": 0.1, "mu_lower_lim": -3,
                       "num_neurons": H, "num_hidden_layers": 2, "dropout": dropout, "dropout_rate": 0.2, "batch_normalization": bn}}}
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, D, generator=gen).clamp(-3, 5).to(dev)
    c = torch.randn(B, C, generator=gen).to(dev)
    eps = torch.randn(S, B, 10, generator=gen).to(dev)
    monkeypatch.setenv("BTVAE_STEP_GRAPH", "0")
    runs = {}
    for padded in (True, False):
        monkeypatch.setattr(E, "PITCHED_ROWS", padded)
        torch.manual_seed(21)
        m = CVAE(input_dim=D, conditioner=Cdim(C), **cfg).to(dev).train()
        m.train_kwargs = {"num_mc_samples": S, "beta": 0.8, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
        opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
        torch.manual_seed(33)                     # the dropout seeds of the steps
        losses = []
        for _ in range(3):
            out = m._train_step(x, c, opt, eps=eps)
            losses.append([out[k].item() for k in ("loss", "elbo", "rll", "kl")])
        m.eval()
        with torch.no_grad():
            fw = m._run_forward(x, c, S, eps=eps)
            ev = m._loss_from_forward(fw, x, 0.8).tolist()
        probe = E._rows_on_lines(4, H, dev)
        assert (probe.stride(0) == 96) == padded
        runs[padded] = (losses, ev, {k: v.detach().clone() for k, v in m.state_dict().items()})
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    assert runs[True][1] == runs[False][1]
    for k, v in runs[False][2].items():
        assert torch.equal(v, runs[True][2][k]), k
    from behavio_temporal_vae_amd import _native as nv
    nv.set_gemm_precision("fp32")


class _FixedRandn:
    """Make ``torch.randn`` return a given tensor once (the module-level API draws its own noise)."""
    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        self._orig = torch.randn
        def randn(*a, **k):
            torch.randn = self._orig
            return self.eps.clone()
        torch.randn = randn

    def __exit__(self, *exc):
        torch.randn = self._orig


@pytest.mark.parametrize("name", ["tiny_dict_2d", "tiny_normal_2d", "tiny_dict_3d"])
def test_module_api_autograd_vs_reference(dev, name):
    """The reference's own training idiom on the module-level API -- forward, loss, ``loss["loss"].backward()`` --
    builds an autograd graph over the HIP kernels: loss terms and every parameter gradient against the reference."""
    g = load_golden(name)
    m = build(g, dev)
    m.train()
    m.prior_params = {k: v.to(dev) for k, v in m.prior_params.items()}
    x, c, eps = T(g["x"]).to(dev), T(g["c"]).to(dev), T(g["train_eps"]).to(dev)
    S = eps.shape[0]
    shaped = eps if c.ndim == 2 else eps.unsqueeze(0)
    with _FixedRandn(shaped):
        xd, zd = m(x, c, num_mc_samples=S)
    assert xd["params"]["mu"].requires_grad and zd["params"]["sigma"].requires_grad
    assert np.allclose(xd["params"]["mu"].detach().cpu().numpy(), g["train_lik_mu"], rtol=1e-4, atol=5e-5)
    loss = m.loss(x, xd["params"], zd["params"], beta=0.7, prior_params=m.prior_params)
    for k in ("loss", "elbo", "rll", "kl"):
        assert abs(loss[k].item() - float(g[f"train_{k}"])) <= 1e-4 * abs(float(g[f"train_{k}"])) + 1e-5, k
    if name == "tiny_dict_3d":
        return                                  # the reference fixture holds gradients for the 2-D cases
    m._grad_arena.zero_()
    loss["loss"].backward()
    for k, p in m.named_parameters():
        ref = g["grad::" + k]
        scale = np.abs(ref).max() + 1e-8
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 5e-4 * scale + 1e-6, (k, err, scale)


def test_module_api_with_stock_optimizer(dev):
    """``torch.optim.Adam`` + ``clip_grad_norm_`` over ``model.parameters()`` with ``zero_grad(set_to_none=True)``:
    gradient buffers come back as arena views, a second backward accumulates, and the trajectory equals the fused
    ``train_core`` path step for step."""
    g = load_golden("tiny_dict_2d")
    ma, mb = build(g, dev), build(g, dev)
    for m in (ma, mb):
        m.train()
        m.prior_params = {k: v.to(dev) for k, v in m.prior_params.items()}
        m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt_a = torch.optim.Adam([{"params": ma.encoder.parameters()}, {"params": ma.decoder.parameters()}], lr=1e-3, weight_decay=1e-5)
    opt_b = mb.get_optimizer(lr=1e-3, weight_decay=1e-5)
    for t in range(g["steps_x"].shape[0]):
        x, c, eps = T(g["steps_x"][t]).to(dev), T(g["steps_c"][t]).to(dev), T(g["steps_eps"][t]).to(dev)
        with _FixedRandn(eps):
            xd, zd = ma(x, c, num_mc_samples=eps.shape[0])
        opt_a.zero_grad(set_to_none=True)
        la = ma.loss(x, xd["params"], zd["params"], beta=0.7, prior_params=ma.prior_params)
        la["loss"].backward()
        assert all(p.grad is not None and p.grad.data_ptr() >= ma._grad_arena.data_ptr() for p in ma.parameters())
        torch.nn.utils.clip_grad_norm_(ma.parameters(), max_norm=1.0)
        opt_a.step()
        lb = mb._train_step(x, c, opt_b, eps=eps)
        assert rel_err([la[k].item() for k in ("loss", "rll", "kl")], [lb[k].item() for k in ("loss", "rll", "kl")]) < 1e-4
        assert rel_err([la[k].item() for k in ("loss", "elbo", "rll", "kl")], g["steps_losses"][t]) < 1e-4
    # accumulation semantics of .backward()
    x, c, eps = T(g["x"]).to(dev), T(g["c"]).to(dev), T(g["train_eps"]).to(dev)
    grads = []
    for reps in (1, 2):
        ma.zero_grad(set_to_none=True)
        for _ in range(reps):
            with _FixedRandn(eps):
                xd, zd = ma(x, c, num_mc_samples=eps.shape[0])
            ma.loss(x, xd["params"], zd["params"], beta=0.7, prior_params=ma.prior_params)["loss"].backward()
        grads.append(ma._grad_arena.clone())
    assert torch.allclose(grads[1], 2 * grads[0], rtol=2e-3, atol=1e-6)


def test_module_api_with_fused_optimizer(dev):
    """The reference's idiom with the model's own optimiser: ``opt = model.get_optimizer()``, forward,
    ``opt.zero_grad(set_to_none=True)``, ``loss.backward()``, clip, ``opt.step()``.  The autograd nodes accumulate
    into the gradient arena, so ``FusedClipAdam.zero_grad`` has to clear it: the trajectory must equal the fused
    ``train_core`` path and the reference over all recorded steps (gradients summed across steps would not)."""
    g = load_golden("tiny_dict_2d")
    ma, mb = build(g, dev), build(g, dev)
    for m in (ma, mb):
        m.train()
        m.prior_params = {k: v.to(dev) for k, v in m.prior_params.items()}
        m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt_a = ma.get_optimizer(lr=1e-3, weight_decay=1e-5)
    opt_b = mb.get_optimizer(lr=1e-3, weight_decay=1e-5)
    for t in range(g["steps_x"].shape[0]):
        x, c, eps = T(g["steps_x"][t]).to(dev), T(g["steps_c"][t]).to(dev), T(g["steps_eps"][t]).to(dev)
        with _FixedRandn(eps):
            xd, zd = ma(x, c, num_mc_samples=eps.shape[0])
        opt_a.zero_grad(set_to_none=True)
        assert all(p.grad is not None for p in ma.parameters()) and float(ma._grad_arena.abs().max()) == 0.0
        la = ma.loss(x, xd["params"], zd["params"], beta=0.7, prior_params=ma.prior_params)
        la["loss"].backward()
        opt_a.step(max_norm=1.0)
        lb = mb._train_step(x, c, opt_b, eps=eps)
        assert rel_err([la[k].item() for k in ("loss", "rll", "kl")], [lb[k].item() for k in ("loss", "rll", "kl")]) < 1e-4, t
        assert rel_err([la[k].item() for k in ("loss", "elbo", "rll", "kl")], g["steps_losses"][t]) < 1e-4, t
    sa, sb = ma.state_dict(), mb.state_dict()
    for k, v in sa.items():
        if v.is_floating_point():
            assert (v - sb[k]).abs().max() <= 5e-4 * (sb[k].abs().max() + 1e-6), k


def test_train_step_replayed_from_hip_graph(dev, monkeypatch):
    """``train_core`` replayed from a captured HIP graph (two eager steps of a shape, then capture + replay): the same
    four reference steps as ``test_train_steps_vs_reference`` -- loss terms at 1e-4, final weights, BatchNorm counters --
    with a learning-rate change between the replays (the optimiser kernel reads its scalars from device memory), and the
    whole trajectory equal to an eager run."""
    from behavio_temporal_vae_amd import vae_models as VM
    g = load_golden("tiny_dict_2d")
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BTVAE_STEP_GRAPH", mode)
        m = build(g, dev)
        m.train()
        m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": 0.7, "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
        opt = m.get_optimizer(lr=1e-3, weight_decay=1e-5)
        losses = []
        n = g["steps_x"].shape[0]
        for t in range(n + 3):
            i = t % n
            if t == n:                       # past the reference's four steps: an LR scheduler steps in
                for pg in opt.param_groups:
                    pg["lr"] = 2.5e-4
            loss = m._train_step(T(g["steps_x"][i]).to(dev), T(g["steps_c"][i]).to(dev), opt, eps=T(g["steps_eps"][i]).to(dev))
            losses.append([loss[k].item() for k in ("loss", "elbo", "rll", "kl")])
            if t < n:
                assert rel_err(losses[-1], g["steps_losses"][t]) < 1e-4, (mode, t, losses[-1], g["steps_losses"][t])
            if t == n - 1:
                sd = m.state_dict()
                for k, v in sd.items():
                    ref = g["final::" + k]
                    if v.is_floating_point():
                        assert np.abs(v.cpu().numpy() - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-6), (mode, k)
                    else:
                        assert int(v) == int(ref), (mode, k)
        graphs = m.__dict__.get("_step_graphs", {})
        assert (len(graphs) == 1) == (mode == "1"), (mode, len(graphs))
        assert opt._ensure().step_count == n + 3 and int(opt._steps[0]) == n + 3
        runs[mode] = (np.array(losses), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    assert rel_err(runs["1"][0], runs["0"][0]) < 1e-5
    for k, v in runs["0"][1].items():
        if v.is_floating_point():
            assert (runs["1"][1][k] - v).abs().max() <= 1e-5 * (v.abs().max() + 1e-6), k
        else:
            assert int(runs["1"][1][k]) == int(v), k


def test_single_row_batch_in_train_mode_raises_like_batchnorm(dev):
    """A one-profile batch in train mode: torch's BatchNorm1d raises ValueError on the encoder's single row, and so
    does this path; eval mode accepts it."""
    g = load_golden("tiny_dict_2d")
    m = build(g, dev).train()
    x, c = T(g["steps_x"][0][:1]).to(dev), T(g["steps_c"][0][:1]).to(dev)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        m.forward(x, c, num_mc_samples=3)
    m.eval()
    with torch.no_grad():
        out, post = m.forward(x, c, num_mc_samples=3)
    assert out["params"]["mu"].shape[:2] == (3, 1) and torch.isfinite(out["params"]["mu"]).all()


class _RandnQueue:
    """Make successive ``torch.randn`` calls return recorded draws (moved to the requested device) in order."""
    def __init__(self, draws):
        self.draws = list(draws)
        self.used = 0

    def __enter__(self):
        self._orig = torch.randn
        def randn(*size, **k):
            shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
            t = self.draws[self.used]
            assert t.numel() == int(np.prod(shape)), (self.used, tuple(t.shape), tuple(shape))
            self.used += 1
            return t.clone().to(k.get("device", "cpu")).reshape(tuple(shape))
        torch.randn = randn
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig


class _Loader:
    """Fixed [x, c] batches with the two attributes ``fit`` / ``validate`` read from a DataLoader."""
    def __init__(self, batches):
        self.batches = batches
        self.dataset = range(sum(x.shape[0] for x, _ in batches))

    def __iter__(self):
        return iter([[x, c] for x, c in self.batches])


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_validate_vs_reference(dev, tag):
    """``CVAE.validate`` (src/vae_models.py:327-337) over three ragged batches, deterministic and sampled (3-D)
    conditions, with the reference's recorded latent draws: dataset-weighted ELBO / RLL / KL at 1e-4."""
    g = load_golden("validate_tiny")
    m = build(g, dev).train()
    n = len(g["sizes"])
    loader = _Loader([(T(g[f"{tag}_x{i}"]), T(g[f"{tag}_c{i}"])) for i in range(n)])
    with _RandnQueue([T(g[f"{tag}_eps{i}"]) for i in range(n)]) as q:
        val = m.validate(loader, num_mc_samples=int(g["S"]), beta=0.7, device=str(dev))
    assert q.used == n and not m.training
    assert rel_err([val["elbo"], val["rll"], val["kl"]], g[f"{tag}_val"]) < 1e-4, (val, g[f"{tag}_val"])


def test_fit_vs_reference(dev, tmp_path, monkeypatch):
    """A short ``fit`` run against the reference's (src/vae_models.py:126-238) on the same batches and latent draws:
    loss terms of every iteration, every validation result and when it happened, the learning-rate change of
    ReduceLROnPlateau(mode=max, threshold_mode=abs), the iteration early stopping ends the run at (the end of the epoch
    in which it was raised), BatchNorm counters and the final weights."""
    g = load_golden("fit_tiny")
    kw = json.loads(str(g["fit_kwargs"]))
    # the replayed noise is handed over as host tensors by a patched torch.randn, which a HIP-graph capture cannot
    # contain: this run stays on the eager path (the graph path has its own test with noise as a captured input)
    monkeypatch.setenv("BTVAE_STEP_GRAPH", "0")
    m = build(g, dev)
    train = _Loader([(T(g[f"train_x{i}"]), T(g[f"train_c{i}"])) for i in range(4)])
    val = _Loader([(T(g[f"val_x{i}"]), T(g[f"val_c{i}"])) for i in range(2)])
    train_log, val_log, lr_log = [], [], []
    tc, vd = m.train_core, m.validate

    def train_core(inputs, optim, **k):
        loss = tc(inputs, optim, **k)
        train_log.append([loss[q].item() for q in ("loss", "elbo", "rll", "kl")])
        return loss

    def validate(*a, **k):
        r = vd(*a, **k)
        val_log.append([r["elbo"], r["rll"], r["kl"], len(train_log)])
        return r

    m.train_core, m.validate = train_core, validate
    step_orig = torch.optim.lr_scheduler.ReduceLROnPlateau.step

    def sched_step(self_, metrics, *a, **k):
        r = step_orig(self_, metrics, *a, **k)
        lr_log.append(self_.optimizer.param_groups[0]["lr"])
        return r

    class W:
        log_dir = str(tmp_path)

    kw["device"] = str(dev)
    torch.optim.lr_scheduler.ReduceLROnPlateau.step = sched_step
    try:
        with _RandnQueue([T(g[f"eps_{i}"]) for i in range(int(g["n_draws"]))]) as q:
            m.fit(train, val, tqdm_func=None, writer=W(), **kw)
    finally:
        torch.optim.lr_scheduler.ReduceLROnPlateau.step = step_orig
        del m.train_core, m.validate
    assert q.used == int(g["n_draws"])
    assert len(train_log) == int(g["iterations"])
    assert rel_err(np.array(train_log), g["train_losses"]) < 1e-4
    assert np.array_equal(np.array(val_log)[:, 3], g["val_log"][:, 3])
    assert rel_err(np.array(val_log)[:, :3], g["val_log"][:, :3]) < 1e-4
    assert np.allclose(lr_log, g["lr_log"], rtol=1e-12)
    assert not m.training
    sd = m.state_dict()
    for k, v in sd.items():
        ref = g["final::" + k]
        if v.is_floating_point():
            assert np.abs(v.cpu().numpy() - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-6), k
        else:
            assert int(v) == int(ref), k


@pytest.mark.parametrize("precision", ["fp32_split2", "fp32_split", "fp32"])
def test_config0_sgd_steps_vs_reference(dev, precision):
    """config0 architecture (H=1000), four ``train_core`` steps of the reference with plain SGD on clipped gradients:
    the update is linear in the gradient, so -- unlike the Adam run of ``test_config0_shape_train_steps``, whose
    sign-like first updates amplify round-off -- every step's loss terms can be held to 1e-4 against the reference's own
    numbers, and each of them depends on the whole backward pass of the steps before."""
    g = load_golden("config0_sgd_steps")
    m = build(g, dev, seed=int(g["seed"])).train()
    m.set_gemm_precision(precision)
    m.train_kwargs = {"num_mc_samples": int(g["S"]), "beta": float(g["beta"]), "gradient_clipping": True, "gradient_clipping_kwargs": {"max_norm": 1.0}}
    opt = torch.optim.SGD(m.parameters(), lr=float(g["lr"]))
    names = json.loads(str(g["grad_keys"]))
    params = dict(m.named_parameters())
    try:
        for t in range(g["steps_x"].shape[0]):
            loss = m._train_step(T(g["steps_x"][t]).to(dev), T(g["steps_c"][t]).to(dev), opt, eps=T(g["steps_eps"][t]).to(dev))
            got = [loss[k].item() for k in ("loss", "elbo", "rll", "kl")]
            assert rel_err(got, g["steps_losses"][t]) < 1e-4, (t, got, g["steps_losses"][t])
            if t == 0:
                # .grad holds the clipped gradient after the step, like the reference's; 64-value slices at 5e-3 of the slice's
                # largest entry (the reference's fp32 gradients are themselves 2e-3 from fp64 ones: norm 2e4 before clipping)
                ref = g["grad0"]
                for i, k in enumerate(names):
                    n = min(64, params[k].numel())
                    mine = params[k].grad.reshape(-1)[:n].cpu().numpy()
                    assert np.abs(mine - ref[i][:n]).max() <= 5e-3 * np.abs(ref[i][:n]).max() + 1e-7, k
    finally:
        m.set_gemm_precision("fp32_split2")
    sd = m.state_dict()
    keys = json.loads(str(g["final_keys"]))
    for i, k in enumerate(keys):
        v = sd[k].double().cpu()
        assert abs(float(v.sum()) - g["final_sums"][i][0]) <= 1e-2 * float(v.abs().sum()) + 1e-6, k   # signed sums of near-zero-mean tensors: percent-level gradient noise does not cancel
        # tensors that started at zero (BatchNorm shifts) hold nothing but four noisy lr * gradient updates: 5 % there
        small = g["final_sums"][i][1] / max(v.numel(), 1) < 1e-6
        assert abs(float((v ** 2).sum()) - g["final_sums"][i][1]) <= (5e-2 if small else 5e-3) * g["final_sums"][i][1] + 1e-9, k
        n = min(64, v.numel())
        # 4 updates of lr * (clipped gradient entries ~1e-4 with percent-level fp32 noise): 3e-6 absolute is that noise
        assert np.abs(v.reshape(-1)[:n].numpy() - g["final_head"][i][:n]).max() <= 2e-3 * np.abs(g["final_head"][i][:n]).max() + 3e-6, k


_FULL = {}


def _full_size_oracle():
    """One config0 train step at the benchmark's size (B=4096, S=16, H=1000) through the fp64 CPU oracle, from
    seed-exact weights: loss terms, gradients, BatchNorm running statistics.  Computed once per session (~1 min)."""
    if _FULL:
        return _FULL
    import contextlib, io
    from oracle import refimpl as R
    from behavio_temporal_vae_amd.vae_models import CVAE
    g = load_golden("config0_sgd_steps")
    cfg = json.loads(str(g["model_cfg"]))
    B, S, D, C, L = 4096, 16, 48, 4, 48
    torch.manual_seed(2112)
    with contextlib.redirect_stdout(io.StringIO()):
        m = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(B, D, generator=gen).add_(1.0).clamp_(-3, 5)
    x[torch.rand(B, D, generator=gen) < 0.1] = -3.0
    ang = torch.rand(B, 2, generator=gen) * (2 * np.pi)
    c = torch.cat([torch.cos(ang), torch.sin(ang)], 1)
    eps = torch.randn(S, B, L, generator=gen)
    spec = R.model_spec(D, C, cfg)
    sd64 = R.cast_state({k: v.clone() for k, v in sd0.items()}, torch.float64)
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    terms, grads, norm = R.train_step(sd64, spec, None, x.double(), c.double(), eps.double(), lr=0.0, weight_decay=0.0, beta=1.0,
                                      max_norm=1.0, optimizer="sgd")
    _FULL.update(cfg=cfg, sd0=sd0, x=x, c=c, eps=eps, terms=terms, grads=grads, norm=norm, after=sd64)
    return _FULL


@pytest.mark.timeout(900)
@pytest.mark.parametrize("precision", ["fp32_split2", "fp32_split2_fp32_activations", "fp32_split", "fp32"])
def test_config0_full_size_step_vs_oracle(dev, precision):
    """The benchmark's own configuration -- config0 model, B=4096, S=16 (65,536 decoder rows), H=1000 -- for one train
    step in the three fp32-accurate GEMM modes against the fp64 oracle on the same seeded weights, data and noise: ELBO /
    RLL / KL at 1e-4, the pre-clip gradient norm, every gradient tensor, and the BatchNorm running statistics the
    forward leaves (statistics over 65,536 rows, the fused BatchNorm-backward epilogues and the split-reduction
    weight gradients at M=65,536 all sit on this path).  The two-piece mode keeps the materialised hidden activations between
    BatchNorm units in the pair format from 65,536 rows on (``engine.TRAIN_PAIRS``); ``fp32_split2_fp32_activations`` is that
    mode without them, held to the same bounds."""
    from behavio_temporal_vae_amd.vae_models import CVAE
    from behavio_temporal_vae_amd import engine as E, _native as nv
    o = _full_size_oracle()
    torch.manual_seed(2112)
    m = CVAE(input_dim=48, conditioner=Cdim(4), **o["cfg"])
    m.load_state_dict(o["sd0"])
    m = m.to(dev).train()
    pairs = precision == "fp32_split2"
    precision = precision.replace("_fp32_activations", "")
    m.set_gemm_precision(precision)
    calls, call0, flag0 = {}, nv.call, E.TRAIN_PAIRS

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return call0(name, *a)
    try:
        E.TRAIN_PAIRS, nv.call = pairs, counting
        fw = m._run_forward(o["x"].to(dev), o["c"].to(dev), 16, eps=o["eps"].to(dev), tape=True)
        out = m._loss_and_backward(fw, o["x"].to(dev), 1.0).tolist()
    finally:
        E.TRAIN_PAIRS, nv.call = flag0, call0
        m.set_gemm_precision("fp32_split2")
    if pairs:
        # the decoder (65,536 rows): the outputs of the trunk's first three units are written as pairs (the last one stays a
        # folded BatchNorm in front of the heads); units two to four read them, forward and in their weight gradients
        assert calls.get("btvae_bn_apply_fwd_pairs", 0) == 3 and calls.get("btvae_linear_relu_stats_fwd_pairs", 0) == 3 \
            and calls.get("btvae_linear_wgrad_pairs", 0) == 3, calls
    else:
        assert "btvae_bn_apply_fwd_pairs" not in calls
    for v, k in zip(out, ("loss", "elbo", "rll", "kl")):
        assert abs(v - o["terms"][k]) <= 1e-4 * abs(o["terms"][k]), (precision, k, v, o["terms"][k])
    gn = float(m._grad_arena.double().pow(2).sum().sqrt())
    assert abs(gn - o["norm"]) <= 1e-3 * o["norm"], (gn, o["norm"])
    # Per-tensor gradient error in the L2 sense.  The yardstick is what fp32 arithmetic itself delivers here: the fp32
    # CPU oracle's gradients are up to 2.3e-3 (L2) / 1.3e-2 (largest entry) away from the fp64 ones at this size, because
    # sums over 65,536 rows cancel to a small fraction of their terms.  Tensors whose exact gradient is zero (a Linear
    # bias in front of a train-mode BatchNorm) are round-off on both sides and are skipped by the norm floor.
    errs = []
    for k, p in m.named_parameters():
        ref = o["grads"][k]
        if float(ref.norm()) < 1e-6 * o["norm"]:
            continue
        errs.append((float((p.grad.double().cpu() - ref).norm() / ref.norm()), k))
    errs.sort(reverse=True)
    assert errs[0][0] <= 6e-3, errs[:5]
    # the square hidden-layer weights: large-norm sums rather than cancellations, held to a third of that
    big = [(e, k) for e, k in errs if k.endswith(".weight") and tuple(dict(m.named_parameters())[k].shape) == (1000, 1000)]
    assert len(big) == 10 and max(big)[0] <= 2e-3, sorted(big, reverse=True)[:4]
    m._flush_bn_counters()
    sd = m.state_dict()
    for k, ref in o["after"].items():
        if "running_" in k:
            got = sd[k].double().cpu()
            assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-7, k
        elif "num_batches_tracked" in k:
            assert int(sd[k]) == int(ref), k


@pytest.mark.parametrize("strict", [True, False])
def test_two_piece_guard_in_a_training_step(dev, strict):
    """A config0-shaped step (H = 1000) in which one profile is 2^20 times the others: the encoder's input, every hidden
    activation behind it (its row dominates) and the head gradients then have regions far below their tensors' maxima,
    where two scaled fp16 pieces keep only an absolute error.  With the guard waited for (strict) the affected products
    run on three exact pieces from the first launch on and the step meets the fp64 oracle at 1e-4; without waiting, the
    first step raises the flags and the second one falls back."""
    import contextlib, io
    from oracle import refimpl as R
    from behavio_temporal_vae_amd.vae_models import CVAE
    from behavio_temporal_vae_amd import engine as E
    g = load_golden("config0_sgd_steps")
    cfg = json.loads(str(g["model_cfg"]))
    B, S, D, C, L = 256, 4, 48, 4, 48
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        m = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(78)
    x = torch.randn(B, D, generator=gen).add_(1.0).clamp_(-3, 5)
    x[0] *= 2.0 ** 20
    ang = torch.rand(B, 2, generator=gen) * (2 * np.pi)
    c = torch.cat([torch.cos(ang), torch.sin(ang)], 1)
    eps = torch.randn(S, B, L, generator=gen)
    spec = R.model_spec(D, C, cfg)
    sd64 = R.cast_state({k: v.clone() for k, v in sd0.items()}, torch.float64)
    terms, grads, norm = R.train_step(sd64, spec, None, x.double(), c.double(), eps.double(), lr=0.0, weight_decay=0.0, beta=1.0,
                                      max_norm=1.0, optimizer="sgd")
    m = m.to(dev).train()
    assert m.gemm_precision == "fp32_split2"
    E.GUARD.reset()
    was, rows_was = E.GUARD.strict, E.TRAIN_PAIRS_MIN_ROWS
    E.GUARD.strict = strict
    E.TRAIN_PAIRS_MIN_ROWS = 0                    # pair-format activations at this size too: a flagged site must leave them
    try:
        xd, cd, ed = x.to(dev), c.to(dev), eps.to(dev)
        fw = m._run_forward(xd, cd, S, eps=ed, tape=True)
        out = m._loss_and_backward(fw, xd, 1.0).tolist()
        torch.cuda.synchronize()
        assert E.GUARD.tripped_sites() > 0
        if strict:
            assert E.GUARD.fallbacks > 0
        else:
            assert E.GUARD.fallbacks == 0
            fw = m._run_forward(xd, cd, S, eps=ed, tape=True)          # the host has seen the flags by now
            out = m._loss_and_backward(fw, xd, 1.0).tolist()
            assert E.GUARD.fallbacks > 0
        E.GUARD.check()                                                 # nothing overflowed
    finally:
        E.GUARD.strict, E.TRAIN_PAIRS_MIN_ROWS = was, rows_was
        E.GUARD.reset()
    for v, k in zip(out, ("loss", "elbo", "rll", "kl")):
        assert abs(v - terms[k]) <= 1e-4 * abs(terms[k]), (strict, k, v, terms[k])
    gn = float(m._grad_arena.double().pow(2).sum().sqrt())
    assert abs(gn - norm) <= 2e-3 * norm, (gn, norm)


def test_two_piece_guard_raises_on_overflow(dev):
    """A maximum slot that understates its tensor (here: overwritten after the producer filled it) overflows fp16 in the
    product that trusts it; the product records non-finite accumulators and ``GUARD.check()`` raises."""
    from behavio_temporal_vae_amd import engine as E, _native as nv
    if nv.get_gemm_precision() != "fp32_split2":
        nv.set_gemm_precision("fp32_split2")
    E.GUARD.reset()
    M, K, N = 256, 1000, 128
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.1
    y = torch.empty(M, N, device=dev)
    part = torch.empty(2, 2, N, device=dev)
    E.MAX_SLOTS.fresh()
    xs, ws = E._amax(x), E._amax(W)
    xs[0] = 1e-3                                  # stale: far below max |x|
    E._operand_maxima(xs, ws, ("test", "overflow"))
    nv.call("btvae_linear_relu_stats_fwd", nv.ptr(x), K, nv.ptr(W), nv.ptr(torch.zeros(N, device=dev)), nv.ptr(y), nv.ptr(part), M, K, N)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="non-finite accumulators"):
        E.GUARD.check()
    E.GUARD.reset()


def test_dict_gauss_beyond_64_dimensions(dev):
    """A dictionary-Gaussian decoder with 96 output coordinates (the reference has no limit on the dimension,
    src/submodels.py:169-179; up to 64 the covariance rows live in registers, beyond that in LDS, one workgroup per row):
    one train step, the importance-weighted score and the imputation conditional against the fp64 oracle at 1e-4."""
    import contextlib, io
    from oracle import refimpl as R
    from behavio_temporal_vae_amd.vae_models import CVAE
    from behavio_temporal_vae_amd import testing_lib, imputation_lib
    D, C, L, B, S, K = 96, 4, 12, 40, 3, 5
    cfg = {"latent_dim": L, "distribution_dict": {
        "posterior": {"dist_type": "normal", "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0, "num_neurons": 72,
                      "num_hidden_layers": 2, "dropout": False, "batch_normalization": True},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 128, "sigma_lim": 1e-6, "marginal_std_lim": 0.05, "mu_lower_lim": -3,
                       "num_neurons": 72, "num_hidden_layers": 2, "dropout": False, "batch_normalization": True}}}
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        m = CVAE(input_dim=D, conditioner=Cdim(C), **cfg)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(B, D, generator=gen).mul_(0.7).add_(0.5)
    c = torch.randn(B, C, generator=gen)
    eps = torch.randn(S, B, L, generator=gen)
    spec = R.model_spec(D, C, cfg)
    sd64 = R.cast_state({k: v.clone() for k, v in sd0.items()}, torch.float64)
    terms, grads, norm = R.train_step(dict(sd64), spec, None, x.double(), c.double(), eps.double(), lr=0.0, weight_decay=0.0, beta=1.0,
                                      max_norm=None, optimizer="sgd")
    m = m.to(dev).train()
    xd, cd = x.to(dev), c.to(dev)
    fw = m._run_forward(xd, cd, S, eps=eps.to(dev), tape=True)
    out = m._loss_and_backward(fw, xd, 1.0).tolist()
    for v, k in zip(out, ("loss", "elbo", "rll", "kl")):
        assert abs(v - terms[k]) <= 1e-4 * abs(terms[k]), (k, v, terms[k])
    gn = float(m._grad_arena.double().pow(2).sum().sqrt())
    assert abs(gn - norm) <= 1e-3 * norm, (gn, norm)
    gU = dict(m.named_parameters())["decoder.SigmaMapper"].grad.double().cpu()
    assert float((gU - grads["decoder.SigmaMapper"]).norm() / grads["decoder.SigmaMapper"].norm()) <= 1e-3
    # eval mode: K-sample score and the conditional of 37 missing coordinates
    m.load_state_dict(sd0)
    m.eval()
    epsk = torch.randn(K, B, L, generator=gen)
    ll, rlls, _ = testing_lib.iw_loglikelihood_device(m, xd, cd, K, mc_sample_batch_size=2, eps=epsk.to(dev), return_rlls=True)
    ll64, rl64, _ = R.iw_loglikelihood(sd64, spec, x.double(), c.double(), epsk.double())
    assert torch.allclose(rlls.cpu().double(), rl64, rtol=1e-4, atol=1e-4 * float(rl64.abs().max()))
    assert torch.allclose(ll.cpu().double(), ll64, rtol=1e-4, atol=1e-4 * float(ll64.abs().max()))
    obs = [i for i in range(D) if i % 5 != 2 and not 60 <= i < 78]
    lik64, _, _ = R.cvae_forward(sd64, spec, x.double(), c.double(), epsk[:1].double(), False)
    mean64, cov64 = R.conditional_params(sd64["decoder.SigmaMapper"], spec.dec, lik64["mu"][0], lik64["sigma"][0], x.double()[:, obs], obs)
    params = {k: v[0].float().to(dev) for k, v in lik64.items()}
    mean, cov = imputation_lib.get_conditional_params(m.decoder, params, xd[:, obs], torch.tensor(obs))
    assert torch.allclose(mean.cpu().double(), mean64, rtol=2e-4, atol=2e-4 * float(mean64.abs().max()))
    assert torch.allclose(cov.cpu().double(), cov64, rtol=2e-4, atol=2e-5 * float(cov64.abs().max()))
