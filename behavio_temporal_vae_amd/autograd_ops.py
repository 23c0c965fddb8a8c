"""``torch.autograd`` nodes over the HIP engine, for callers that write their own training loop with the
module-level API (``x_dict, z_dict = model(x, c, S); model.loss(...)["loss"].backward()``), as the reference's
``train_core`` itself does (src/vae_models.py:317-325).

Two nodes cover everything that is not cheap element-wise arithmetic:

* ``NetFunction`` — a whole distribution network (trunk + heads + output non-linearities).  Its backward runs the
  engine's backward pass; parameter gradients are ACCUMULATED straight into the ``.grad`` views of the model's
  gradient arena (autograd receives ``None`` for them), the input gradient flows on through autograd.
* ``DictLogLik`` — the dictionary-Gaussian multivariate-normal log-density; backward = the analytic gradient
  kernels (per-row upstream gradients), dictionary gradient accumulated into ``SigmaMapper.grad``.

``model.train_core`` does not go through here: it uses the fused path that never builds a graph.
"""
from __future__ import annotations

import torch

from . import _native as nv
from . import engine as E


def _ensure_grad_buffers(module: torch.nn.Module):
    """Parameters whose ``.grad`` was set to None (``zero_grad(set_to_none=True)`` of a stock optimizer) get their
    view of the owning model's gradient arena back, zeroed."""
    owner = getattr(module, "_arena_owner", None)
    owner = owner() if owner is not None else None
    if owner is None:
        raise RuntimeError("this network is not attached to a model arena; build it through VAE/CVAE")
    owner._reattach_grads()


class NetFunction(torch.autograd.Function):
    """(dist module, input rows, *its parameters) -> head values in ``head_specs`` order."""

    @staticmethod
    def forward(ctx, dist, x, *params):
        plan = dist.plan()
        train = dist.training
        tape = E.NetTape() if train else None
        out, _ = E.net_forward(plan, x, train, tape, dist.head_specs())
        if train:
            plan.nbt_pending = getattr(plan, "nbt_pending", 0) + 1
        ctx.dist, ctx.tape, ctx.names = dist, tape, list(out)
        ctx.set_materialize_grads(False)
        return tuple(out[n] for n in ctx.names)

    @staticmethod
    def backward(ctx, *gouts):
        if ctx.tape is None:
            raise RuntimeError("backward through a network in eval mode is not supported (BatchNorm is folded); call model.train()")
        _ensure_grad_buffers(ctx.dist)
        draw = {}
        for name, g in zip(ctx.names, gouts):
            dfac = ctx.tape.dfac[name]
            draw[name] = (g.contiguous() * dfac) if g is not None else torch.zeros_like(dfac)
        dx = E.net_backward(ctx.dist.plan(), ctx.tape, draw, need_input_grad=ctx.needs_input_grad[1], accumulate=True)
        ctx.tape = None
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


def apply_net(dist, x: torch.Tensor):
    """Run ``dist``'s network on rows ``x`` with an autograd node attached; returns {name: value}."""
    params = [p for p in dist.parameterizer.parameters()]
    vals = NetFunction.apply(dist, x, *params)
    return dict(zip(dist.head_specs().keys(), vals))


class DictLogLik(torch.autograd.Function):
    """(decoder, x rows [B,D], mu rows [R,D], sigma rows [R,V], SigmaMapper) -> log N(x[r % B]; mu_r, Sigma_r) [R]."""

    @staticmethod
    def forward(ctx, dec, x, mu, sigma, U):
        tab = dec.tables()
        sq = (sigma * sigma).contiguous()
        logp = E.dict_loglik_fwd(tab, x, mu, sq)
        ctx.dec, ctx.tab = dec, tab
        ctx.save_for_backward(x, mu, sigma, sq)
        return logp

    @staticmethod
    def backward(ctx, g):
        x, mu, sigma, sq = ctx.saved_tensors
        dec = ctx.dec
        _ensure_grad_buffers(dec)
        ones_mu, ones_sg = torch.ones_like(mu), torch.ones_like(sigma)
        _, dmu, dsig = E.dict_loglik_fwd_bwd(ctx.tab, x, mu, sigma, sq, ones_mu, ones_sg, 0.0, dec.SigmaMapper.grad,
                                             accumulate=True, g_rows=g.contiguous())
        return None, None, dmu, dsig, None


def dict_log_likelihood(dec, x_rows, mu_rows, sigma_rows):
    nv.ensure_device(mu_rows.device)
    return DictLogLik.apply(dec, x_rows, mu_rows, sigma_rows, dec.SigmaMapper)
