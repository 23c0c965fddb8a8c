"""Conditional VAE with the reference's model API (``src/vae_models.py``), computed by HIP kernels.

Drop-in surface: ``CVAE(input_dim, conditioner, latent_dim, distribution_dict)``, ``forward``, ``loss``,
``train_core``, ``validate``, ``sample``, ``reconstruct``, ``fit``, ``get_optimizer``, ``save``/``load``
and the ``prior_params`` / ``model_kwargs`` / ``train_kwargs`` / ``log_dir`` attributes; checkpoints use the
reference's state-dict key names.  Parameters live in one flat fp32 arena (encoder then decoder) with a
parallel gradient arena, so the optimiser step and the data-parallel gradient all-reduce are single
flat-buffer operations.  There is no CPU compute path: tensors must be on an MI355X.
"""
from __future__ import annotations

import datetime
import gc
import json
import os
import weakref
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _native as nv
from . import engine as E
from .submodels import DictionaryGaussian, GaussianNN, get_distribution_model, get_prior_params
from .utils import EarlyStopping

try:  # tensorboard is optional: the container may not ship it
    from torch.utils.tensorboard import SummaryWriter  # type: ignore
except Exception:  # pragma: no cover
    SummaryWriter = None


# Unless a model asks otherwise (``gemm_precision=`` in its config, or BTVAE_GEMM_PRECISION), the Linear products run on two
# fp16 pieces of each operand value scaled per tensor ("fp32_split2": three matrix instructions per product; error against
# fp64 at the fp32 MFMA instruction's level while a tensor's values lie within ~2^17 of its maximum).  The engine's guard
# watches that condition per product and routes a product whose operand breaks it to "fp32_split" (three bf16 pieces: the
# fp32 values exactly, six matrix instructions), which is also what products without tracked maxima run on.
DEFAULT_GEMM_PRECISION = os.environ.get("BTVAE_GEMM_PRECISION", "fp32_split2")


class ScalarLogWriter:
    """Minimal stand-in for ``SummaryWriter`` when tensorboard is not installed: owns ``log_dir`` and
    appends scalars to ``scalars.jsonl``."""

    def __init__(self, log_dir=None):
        if log_dir is None:
            log_dir = os.path.join("runs", datetime.datetime.now().strftime("%b%d_%H-%M-%S"))
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir

    def add_scalars(self, tag, values, step):
        with open(os.path.join(self.log_dir, "scalars.jsonl"), "a") as f:
            f.write(json.dumps({"tag": tag, "step": int(step), **{k: float(v) for k, v in values.items()}}) + "\n")


def make_writer(log_dir=None):
    if SummaryWriter is not None:
        return SummaryWriter(log_dir=log_dir) if log_dir is not None else SummaryWriter()
    return ScalarLogWriter(log_dir)


@dataclass
class _Fwd:
    """Everything one forward pass produced (device tensors)."""
    S: int
    B: int
    three_d: bool
    post: Dict[str, torch.Tensor]        # [R, L]
    post_dfac: Dict[str, torch.Tensor]
    eps: torch.Tensor                    # [S*B, L]
    zc: torch.Tensor                     # [S*B, L+C]
    klrow: torch.Tensor                  # [R]
    lik: Dict[str, torch.Tensor]         # [S*B, .]
    lik_sq: Dict[str, torch.Tensor]
    enc_tape: Optional[E.NetTape]
    dec_tape: Optional[E.NetTape]


class FusedClipAdam(torch.optim.Optimizer):
    """Adam with L2 weight decay over the model's flat arenas, fused with global-norm clipping.
    Keeps torch's two parameter groups (encoder, decoder) so LR schedulers work unchanged
    (reference: src/vae_models.py:102-103).  Step count, bias corrections and hyper-parameters live in device memory
    (``btvae_adam_tick`` / ``btvae_clip_adam_step_dev``), so a step captured in a HIP graph keeps following them."""

    def __init__(self, model, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8):
        groups = [{"params": list(model.encoder.parameters())}, {"params": list(model.decoder.parameters())}]
        super().__init__(groups, dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))
        self.model = model
        self._state_dev = None
        self._adam: Optional[E.ClipAdam] = None
        self._hyper = None            # [2, 8] device floats, one row per span
        self._steps = None            # [1] device int64 step counter
        self._hyper_cache = [None, None]

    def zero_grad(self, set_to_none: bool = True):
        """Zero the gradient arena in both modes and keep the parameters' ``.grad`` views alive.  The fused
        ``train_core`` overwrites gradients and does not need this, but the module-level idiom of the reference
        (``forward`` -> ``zero_grad(set_to_none=True)`` -> ``loss.backward()`` -> ``step()``, src/vae_models.py:319-324)
        accumulates into the arena through autograd, so the buffers must start at zero: one 42 MB memset."""
        self.model._grad_arena.zero_()

    def _ensure(self):
        arena = self.model._arena
        if self._adam is None or self._state_dev != arena.device or self._adam.m.numel() != arena.numel():
            old = self._adam
            self._adam = E.ClipAdam(arena.numel(), arena.device)
            if old is not None and old.m.numel() == arena.numel():
                self._adam.m.copy_(old.m); self._adam.v.copy_(old.v); self._adam.step_count = old.step_count
            self._state_dev = arena.device
            self._hyper = torch.zeros(2, 8, device=arena.device)
            self._steps = torch.full((1,), self._adam.step_count, dtype=torch.int64, device=arena.device)
            self._hyper_cache = [None, None]
        return self._adam

    def _spans(self):
        m = self.model
        g0, g1 = self.param_groups
        same = all(g0[k] == g1[k] for k in ("lr", "weight_decay", "betas", "eps"))
        n = m._arena.numel()
        return [(0, n, g0)] if same else [(0, m._enc_span, g0), (m._enc_span, n - m._enc_span, g1)]

    def refresh_hyper(self, max_norm: Optional[float] = None):
        """Bring the device copies of (lr, betas, eps, weight decay, clipping norm) up to date with the parameter
        groups; a host-to-device copy only when something changed (an LR scheduler).  Returns the spans."""
        self._ensure()
        spans = self._spans()
        for i, (_, _, g) in enumerate(spans):
            vals = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                    float(max_norm) if max_norm is not None else -1.0)
            if self._hyper_cache[i] != vals:
                self._hyper[i, :6].copy_(torch.tensor(vals, dtype=torch.float32))
                self._hyper_cache[i] = vals
        return spans

    @torch.no_grad()
    def step(self, closure=None, max_norm: Optional[float] = None):
        m = self.model
        adam = self._ensure()
        spans = self.refresh_hyper(max_norm)
        n = m._arena.numel()
        adam.step_count += 1
        nv.call("btvae_grad_sumsq", nv.ptr(m._grad_arena), n, nv.ptr(adam.partials))
        nv.call("btvae_adam_tick", nv.ptr(self._steps), nv.ptr(self._hyper), len(spans))
        for i, (off, cnt, _) in enumerate(spans):
            nv.call("btvae_clip_adam_step_dev", nv.ptr(m._arena[off:]), nv.ptr(m._grad_arena[off:]), nv.ptr(adam.m[off:]), nv.ptr(adam.v[off:]),
                    cnt, nv.ptr(adam.partials), nv.ptr(self._hyper[i]), nv.ptr(adam.norm))
        return None

    @property
    def last_grad_norm(self) -> torch.Tensor:
        return self._ensure().norm


STEP_GRAPH_MAX_ROWS = 32768     # S*B up to which train_core is replayed from a HIP graph (launch-bound below, kernel-bound above)
STEP_GRAPH_WARMUP = 2           # eager steps of one shape before it is captured


class _StepGraph:
    """One ``train_core`` step (forward, ELBO, backward, clip + Adam) captured as a HIP graph.

    The step is ~160 short launches; at the batch size config0.json ships (1024) the host cannot issue them as fast as
    the GPU retires them.  Capturing records the launches once -- every device buffer they touch comes from the graph's
    own memory pool or is persistent (arenas, optimiser state, workspaces, which the object keeps alive) -- and each
    later step is one graph launch after copying the batch into the captured input buffers.  The latent noise comes
    from torch's CUDA generator, which advances its Philox offset per replay; step count, bias corrections and learning
    rate are read from device memory by the optimiser kernel."""

    def __init__(self, model, optim, x, c, eps=None):
        self.x, self.c = x.clone(), c.clone()
        self.eps = eps.clone() if eps is not None else None       # injected noise (tests): a captured input like x and c
        self.keep = (list(E.WS.bufs.values()), optim, optim._ensure(), optim._hyper, optim._steps)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model._train_step_eager(self.x, self.c, optim, eps=self.eps)
        self.keep = self.keep + (list(E.WS.bufs.values()),)

    def run(self, x, c, eps=None):
        self.x.copy_(x)
        self.c.copy_(c)
        if self.eps is not None:
            self.eps.copy_(eps.reshape(self.eps.shape))
        self.graph.replay()
        return self.out.clone()


class VAE(torch.nn.Module):
    """Shared machinery; the conditional model below is the one the shipped configs build."""

    def __init__(self, input_dim=None, latent_dim=10, distribution_dict={}, **_):
        super().__init__()
        self.model_kwargs = {"latent_dim": latent_dim, "distribution_dict": distribution_dict, "_": _}
        self.latent_dim, self.distribution_dict = latent_dim, distribution_dict
        self.input_dim = input_dim
        self.condition_dim = 0
        # same construction (and RNG consumption) order as the reference: src/vae_models.py:29-34
        self.encoder = get_distribution_model(input_dim=input_dim, output_dim=latent_dim, **distribution_dict["posterior"])
        self.decoder = get_distribution_model(input_dim=latent_dim, output_dim=input_dim, **distribution_dict["likelihood"])
        self.prior_params = get_prior_params(distribution_dict["posterior"]["dist_type"], latent_dim)
        self.num_parameters = self.encoder._num_parameters() + self.decoder._num_parameters()
        self.train_kwargs = {}
        self.log_dir = None
        self._arena = None
        self._grad_arena = None
        self._dp_world = 1
        self._dp_active = False
        self._bn_sync = None
        # operand precision of the Linear products: "fp32_split2" (default: two per-tensor-scaled fp16 pieces per value, 3
        # matrix instructions per product, error against fp64 at the fp32 MFMA instruction's level), "fp32_split" (three bf16
        # pieces: the fp32 values exactly, 6 matrix instructions), "fp32" (the fp32 MFMA instruction) or "bf16"; an extra keyword of the model config, swallowed by the reference's
        # constructors like any other unknown key
        self.gemm_precision = DEFAULT_GEMM_PRECISION
        self.set_gemm_precision(_.get("gemm_precision", DEFAULT_GEMM_PRECISION))
        self._finish_init()

    def set_gemm_precision(self, name: str):
        if name not in nv.GEMM_PRECISIONS:
            raise ValueError(f"unknown gemm_precision {name!r}; choose from {sorted(nv.GEMM_PRECISIONS)}")
        self.gemm_precision = name

    # ------------------------------------------------------------------ arenas
    def _finish_init(self):
        if not isinstance(self.encoder, GaussianNN) or not self.encoder.learn_sigma:
            raise NotImplementedError("the HIP path needs a Gaussian posterior with learn_sigma=True")
        self._build_arena()

    def _arena_params(self):
        return list(self.encoder.parameters()) + list(self.decoder.parameters())

    def _build_arena(self):
        """Re-home every parameter as a view of one flat buffer (16-byte aligned starts) and give it a
        ``.grad`` view of a parallel gradient buffer."""
        params = self._arena_params()
        dev = params[0].device
        offs, off = [], 0
        n_enc = len(list(self.encoder.parameters()))
        enc_span = 0
        for i, p in enumerate(params):
            if i == n_enc:
                enc_span = off
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        arena = torch.zeros(off, device=dev, dtype=torch.float32)
        grads = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(params, offs):
                arena[o:o + p.numel()].copy_(p.detach().reshape(-1).float())
                p.data = arena[o:o + p.numel()].view(p.shape)
                p.grad = grads[o:o + p.numel()].view(p.shape)
        self._arena, self._grad_arena, self._enc_span = arena, grads, enc_span
        self._offsets = offs
        # per network: (start, trunk start, trunk end, end) of its slice of the arenas; the trunk's parameters are
        # contiguous, everything else of the network (heads, dictionary) has its gradient ready before the trunk's
        self._net_spans = {}
        for name, net, lo, hi in (("encoder", self.encoder, 0, enc_span), ("decoder", self.decoder, enc_span, off)):
            trunk = {id(q) for q in net.parameterizer.block_dict["input"].parameters()}
            t_offs = [(o, o + (q.numel() + 3) // 4 * 4) for q, o in zip(params, offs) if id(q) in trunk]
            t0, t1 = min(a for a, _ in t_offs), max(b for _, b in t_offs)
            if sum(b - a for a, b in t_offs) != t1 - t0:
                t0, t1 = lo, hi           # trunk not contiguous in parameters() order: reduce the network as one span
            self._net_spans[name] = (lo, t0, t1, hi)
        ref = weakref.ref(self)
        for net in (self.encoder, self.decoder):
            object.__setattr__(net, "_arena_owner", ref)

    def _reattach_grads(self):
        """Give parameters whose ``.grad`` is None (a stock optimizer's ``zero_grad(set_to_none=True)``) their
        view of the gradient arena back, zeroed."""
        for p, o in zip(self._arena_params(), self._offsets):
            if p.grad is None:
                seg = self._grad_arena[o:o + p.numel()]
                seg.zero_()
                p.grad = seg.view(p.shape)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if self._arena is not None:
            p0 = self._arena_params()[0]
            if p0.device != self._arena.device or p0.data_ptr() != self._arena.data_ptr():
                self._build_arena()
        return out

    def _check_arena(self):
        p0 = self._arena_params()[0]
        if p0.data_ptr() != self._arena.data_ptr() or p0.grad is None:
            self._build_arena()
        nv.ensure_device(self._arena.device)
        self._sync_precision()

    def _sync_precision(self):
        if nv.get_gemm_precision() != self.gemm_precision:
            nv.set_gemm_precision(self.gemm_precision)

    def _flush_bn_counters(self):
        for net in (self.encoder, self.decoder):
            plan = net.plan()
            pend = getattr(plan, "nbt_pending", 0)
            if pend:
                for bn in plan.bn_modules():
                    if bn.num_batches_tracked is not None:
                        bn.num_batches_tracked += pend
                plan.nbt_pending = 0

    def state_dict(self, *a, **k):
        self._flush_bn_counters()
        return super().state_dict(*a, **k)

    def __getstate__(self):
        # captured step graphs hold device resources of this process: they are rebuilt on demand, never copied or pickled
        state = dict(self.__dict__)
        for k in ("_step_graphs", "_step_seen", "_step_graph_failed"):
            state.pop(k, None)
        return state

    # ------------------------------------------------------------------ forward
    def _prep(self, t: torch.Tensor, name: str) -> torch.Tensor:
        if t.device != self._arena.device:
            raise RuntimeError(f"{name} is on {t.device} but the model is on {self._arena.device}")
        return nv.f32(t, name)

    def _run_forward(self, x, c, S: int, eps=None, tape: bool = False) -> _Fwd:
        """Encoder -> reparameterise (+KL) -> decoder.  ``c`` is None (VAE), [B,C] or [B,S,C]."""
        self._check_arena()
        x = self._prep(x, "inputs")
        B = x.shape[0]
        train = self.training
        if tape and not train:
            raise RuntimeError("a training step needs the model in train mode (call model.train())")
        three_d = c is not None and c.ndim == 3
        E.MAX_SLOTS.fresh()          # every slot this pass uses is zeroed inside it (a captured pass must not inherit slots)
        if c is not None:
            c = self._prep(c, "conditions")
            if c.ndim == 3:
                if c.shape[1] != S:
                    raise ValueError("Condition samples must be equal to number of posterior MC samples.")
                enc_in = E.measure_once(torch.cat((x.unsqueeze(0).expand(S, B, x.shape[1]), c.transpose(1, 0)), dim=-1).reshape(S * B, -1).contiguous())
            elif c.ndim == 2:
                enc_in = E.measure_once(torch.cat((x, c), dim=-1))      # a fresh tensor, read by the forward product and the weight gradient
            else:
                raise ValueError("Invalid shape for conditions. Must be 2D or 3D tensor.")
        else:
            enc_in = x
        enc_tape = E.NetTape() if tape else None
        dec_tape = E.NetTape() if tape else None
        post, _ = E.net_forward(self.encoder.plan(), enc_in.contiguous(), train, enc_tape, self.encoder.head_specs())
        L = self.latent_dim
        if eps is None:
            # same draw the reference makes on this device (src/submodels.py:109)
            R = post["mu"].shape[0]
            eps = torch.randn((1 if three_d else S, R, L), device=x.device)
        eps = self._prep(eps, "eps").reshape(-1, L)
        zc, klrow, eps = E.reparam_kl(post["mu"], post["sigma"], eps, c, S, three_d, B)
        specs = self.decoder.head_specs()
        lik, lik_sq = E.net_forward(self.decoder.plan(), zc, train, dec_tape, specs,
                                    want_sq=("sigma",) if isinstance(self.decoder, DictionaryGaussian) else ())
        if not isinstance(self.decoder, DictionaryGaussian) and "sigma" not in lik:
            lik["sigma"] = torch.full_like(lik["mu"], float(self.decoder.sigma_fixed))
        if train:
            for net in (self.encoder, self.decoder):
                plan = net.plan()
                plan.nbt_pending = getattr(plan, "nbt_pending", 0) + 1
        return _Fwd(S, B, three_d, post, enc_tape.dfac if tape else {}, eps, zc, klrow, lik, lik_sq, enc_tape, dec_tape)

    def _pack_outputs(self, fw: _Fwd):
        S, B, L = fw.S, fw.B, self.latent_dim
        lik = {k: v.view(S, B, v.shape[-1]) for k, v in fw.lik.items()}
        if fw.three_d:
            post = {k: v.view(S, B, L) for k, v in fw.post.items()}
        else:
            post = dict(fw.post)
        z = fw.zc.view(S, B, -1)[..., :L]
        return {"params": lik}, {"params": post, "samples": z}

    def _graph_mode(self) -> bool:
        """Module-level ``forward``/``loss`` build an autograd graph only in train mode under grad mode."""
        return torch.is_grad_enabled() and self.training

    def _forward_graph(self, x, c, S: int):
        """``forward`` with autograd nodes attached (src/vae_models.py:260-280): encoder network, z = mu + sigma*eps,
        decoder network.  Same kernels as the fused path; the glue between them is ordinary torch arithmetic."""
        self._check_arena()
        x = self._prep(x, "inputs")
        B, L = x.shape[0], self.latent_dim
        if c is None:
            post = self.encoder(x)
            z = self.encoder.rsample(post, num_samples=S)
            dec_in = z
        elif c.ndim == 2:
            c = self._prep(c, "conditions")
            post = self.encoder(torch.cat((x, c), dim=-1))
            z = self.encoder.rsample(post, num_samples=S)
            dec_in = torch.cat((z, c.unsqueeze(0).expand(S, B, c.shape[-1])), dim=2)
        elif c.ndim == 3:
            c = self._prep(c, "conditions")
            if c.shape[1] != S:
                raise ValueError("Condition samples must be equal to number of posterior MC samples.")
            ct = c.transpose(1, 0)
            post = self.encoder(torch.cat((x.unsqueeze(0).expand(S, B, x.shape[1]), ct), dim=-1))
            post = {k: v.view(S, B, -1) for k, v in post.items()}
            z = self.encoder.rsample(post, num_samples=1).squeeze(0)
            dec_in = torch.cat((z, ct), dim=-1)
        else:
            raise ValueError("Invalid shape for conditions. Must be 2D or 3D tensor.")
        lik = self.decoder(dec_in.reshape(S * B, -1).contiguous())
        lik = {k: v.view(S, B, -1) for k, v in lik.items()}
        return {"params": lik}, {"params": post, "samples": z}

    def forward(self, inputs, num_mc_samples=1):
        if self._graph_mode():
            return self._forward_graph(inputs, None, num_mc_samples)
        fw = self._run_forward(inputs, None, num_mc_samples)
        return self._pack_outputs(fw)

    @torch.no_grad()
    def reconstruct(self, inputs, num_mc_samples=1):
        return self.forward(inputs, num_mc_samples=num_mc_samples)

    # ------------------------------------------------------------------ loss
    def reconstruction_loglikelihood(self, x, likelihood_params):
        return self.decoder.log_likelihood(x, likelihood_params)

    def kl_divergence(self, posterior_params, prior_params=None):
        if prior_params is None:
            prior_params = self.prior_params
        return self.encoder.kl_divergence(posterior_params, prior_params=prior_params).sum(dim=-1)

    def loss(self, x, likelihood_params, posterior_params, beta=1.0, prior_params=None):
        """ELBO terms as 0-dim device tensors (reference: src/vae_models.py:70-74)."""
        if torch.is_grad_enabled() and any(t.requires_grad for d in (likelihood_params, posterior_params) for t in d.values()):
            if prior_params is not None:
                prior_params = {k: (v.to(x.device) if torch.is_tensor(v) else v) for k, v in prior_params.items()}
            rll = self.reconstruction_loglikelihood(x, likelihood_params).mean()
            kl = self.kl_divergence(posterior_params, prior_params=prior_params).mean()
            return {"loss": -(rll - beta * kl), "elbo": rll - kl, "rll": rll, "kl": kl}
        rows = self.reconstruction_loglikelihood(x, likelihood_params).reshape(-1).contiguous()
        kl = self.kl_divergence(posterior_params, prior_params=prior_params).reshape(-1).contiguous()
        out = E.elbo_scalars(rows, kl, beta)
        return {"loss": out[0], "elbo": out[1], "rll": out[2], "kl": out[3]}

    def _loglik_rows(self, fw: _Fwd, x):
        if isinstance(self.decoder, DictionaryGaussian):
            return E.dict_loglik_fwd(self.decoder.tables(), x, fw.lik["mu"], fw.lik_sq["sigma"])
        return E.diag_loglik(x, fw.lik["mu"], fw.lik["sigma"])[0]

    def _loss_from_forward(self, fw: _Fwd, x, beta: float):
        return E.elbo_scalars(self._loglik_rows(fw, x), fw.klrow, beta)

    def _loss_and_backward(self, fw: _Fwd, x, beta: float) -> torch.Tensor:
        """ELBO scalars plus all parameter gradients of ``loss = -(rll - beta*kl)`` into the gradient arena."""
        x = self._prep(x, "inputs")
        self._sync_precision()
        rows = fw.S * fw.B
        g = -1.0 / rows
        dec = self.decoder
        if isinstance(dec, DictionaryGaussian):
            tab = dec.tables()
            logp, dmu, dsig = E.dict_loglik_fwd_bwd(tab, x, fw.lik["mu"], fw.lik["sigma"], fw.lik_sq["sigma"], fw.dec_tape.dfac["mu"],
                                                    fw.dec_tape.dfac["sigma"], g, dec.SigmaMapper.grad)
            draw = {"mu": dmu, "sigma": dsig}
        else:
            ones = None
            if "sigma" not in fw.dec_tape.dfac:
                ones = torch.zeros_like(fw.lik["mu"])
            logp, dmu, dsig = E.diag_loglik(x, fw.lik["mu"], fw.lik["sigma"], fw.dec_tape.dfac["mu"],
                                            fw.dec_tape.dfac.get("sigma", ones), g)
            draw = {"mu": dmu}
            if "sigma" in fw.dec_tape.dfac:
                draw["sigma"] = dsig
        out = E.elbo_scalars(logp, fw.klrow, beta)
        dzc = E.net_backward(dec.plan(), fw.dec_tape, draw, need_input_grad=True, after_heads=lambda: self._grads_ready("decoder", "heads"))
        self._grads_ready("decoder", "trunk")
        R = fw.post["mu"].shape[0]
        dm, ds = E.reparam_kl_bwd(dzc, fw.eps, fw.post["mu"], fw.post["sigma"], fw.post_dfac["mu"], fw.post_dfac["sigma"],
                                  fw.S, fw.three_d, self.condition_dim, beta / R)
        E.net_backward(self.encoder.plan(), fw.enc_tape, {"mu": dm, "sigma": ds}, need_input_grad=False,
                       after_heads=lambda: self._grads_ready("encoder", "heads"))
        self._grads_ready("encoder", "trunk")
        return out

    # ------------------------------------------------------------------ data parallel hooks
    def enable_data_parallel(self, process_group=None, sync_batchnorm=False, force_collectives=False):
        """Average gradients (and the reported loss terms) over the ranks of ``process_group`` (RCCL) in
        every ``train_core``.  Decoder gradients are reduced while the encoder backward still runs.
        ``sync_batchnorm=True`` also makes the BatchNorm statistics span the ranks (equal rows per rank), so N ranks
        on 1/N of a batch each reproduce the single-device step on the whole batch; the default keeps them per
        rank, as ``DistributedDataParallel`` over the reference's modules would.  ``force_collectives=True`` issues
        every collective in a group of one rank as well, where each is an identity: the step must then equal the
        plain single-device step bit for bit, which is how the RCCL call sequence is tested on one GPU."""
        from .parallel import BatchNormSync, GradSync
        self._dp_sync = GradSync(process_group, force=force_collectives)
        self._dp_world = self._dp_sync.world
        self._dp_active = self._dp_sync.active
        bn = BatchNormSync(process_group, force=force_collectives) if sync_batchnorm else None
        self._bn_sync = bn if (bn is not None and bn.active) else None      # installed in the engine for this model's steps only

    def _grads_ready(self, which: str, part: str):
        """Start the sum all-reduce of the gradients that have just been enqueued: a network's heads (and the decoder's
        dictionary) once their backward is issued, its trunk after the trunk's."""
        if not self._dp_active:
            return
        lo, t0, t1, hi = self._net_spans[which]
        if part == "trunk":
            self._dp_sync.launch(self._grad_arena[t0:t1])
        else:
            for a, b in ((lo, t0), (t1, hi)):
                if b > a:
                    self._dp_sync.launch(self._grad_arena[a:b])

    def sync_gradients(self):
        """Data parallelism with the module-level idiom of the reference (forward, loss, ``loss.backward()``, optimiser
        step: src/vae_models.py:319-324): call between ``backward()`` and the optimiser step.  Averages the whole gradient
        arena over the ranks of ``enable_data_parallel`` (one all-reduce; nothing overlaps, autograd has finished).
        ``train_core`` reduces its own gradients and does not need this; without data parallelism it does nothing."""
        if not self._dp_active:
            return
        self._dp_sync.launch(self._grad_arena)
        self._dp_sync.finish()
        nv.call("btvae_scale", nv.ptr(self._grad_arena), self._grad_arena.numel(), 1.0 / self._dp_world)

    def _dp_finish(self, scalars: torch.Tensor):
        if not self._dp_active:
            return scalars
        self._dp_sync.finish()
        nv.call("btvae_scale", nv.ptr(self._grad_arena), self._grad_arena.numel(), 1.0 / self._dp_world)
        self._dp_sync.sum_(scalars)
        nv.call("btvae_scale", nv.ptr(scalars), 4, 1.0 / self._dp_world)
        return scalars

    # ------------------------------------------------------------------ training
    def get_optimizer(self, lr=1e-3, weight_decay=1e-4):
        self._check_arena()
        return FusedClipAdam(self, lr=lr, weight_decay=weight_decay)

    def _optim_step(self, optim):
        tk = self.train_kwargs
        clip = tk.get("gradient_clipping", False)
        ck = dict(tk.get("gradient_clipping_kwargs", {"max_norm": 1.0})) if clip else {}
        if isinstance(optim, FusedClipAdam):
            simple = set(ck) <= {"max_norm"} or (set(ck) <= {"max_norm", "norm_type"} and float(ck.get("norm_type", 2.0)) == 2.0)
            if clip and not simple:
                torch.nn.utils.clip_grad_norm_(self.parameters(), **ck)
                optim.step()
            else:
                optim.step(max_norm=ck.get("max_norm") if clip else None)
        else:
            if clip:
                torch.nn.utils.clip_grad_norm_(self.parameters(), **ck)
            optim.step()

    def _train_step_eager(self, x, c, optim, eps=None):
        prev, E.BN_SYNC = E.BN_SYNC, (self._bn_sync if self._bn_sync is not None else E.BN_SYNC)
        try:
            fw = self._run_forward(x, c, self.train_kwargs["num_mc_samples"], eps=eps, tape=True)
            out = self._loss_and_backward(fw, x, self.train_kwargs["beta"])
        finally:
            E.BN_SYNC = prev
        out = self._dp_finish(out)
        self._optim_step(optim)
        return out

    def _graph_key(self, x, c, optim, eps):
        """Key of the captured step this call could replay, or None when the call has to run eagerly:
        data parallelism, dropout (its seeds are drawn on the host), sampled 3-D conditions, per-launch event timing, a
        stock torch optimiser, clipping options the fused step does not cover, or more rows than the launch-bound regime."""
        mode = os.environ.get("BTVAE_STEP_GRAPH", "auto")
        if mode == "0" or self._dp_active or self._bn_sync is not None or E.BN_SYNC is not None or not self.training:
            return None
        if eps is not None and (eps.dtype != torch.float32 or eps.device != x.device):
            return None
        if not isinstance(optim, FusedClipAdam) or c is None or c.ndim != 2 or x.dtype != torch.float32 or c.dtype != torch.float32:
            return None
        if not x.is_cuda or x.device != self._arena.device or c.device != x.device:
            return None
        tk = self.train_kwargs
        clip = tk.get("gradient_clipping", False)
        ck = dict(tk.get("gradient_clipping_kwargs", {"max_norm": 1.0})) if clip else {}
        if clip and not (set(ck) <= {"max_norm"} or (set(ck) <= {"max_norm", "norm_type"} and float(ck.get("norm_type", 2.0)) == 2.0)):
            return None
        S = int(tk["num_mc_samples"])
        if mode != "1" and S * x.shape[0] > STEP_GRAPH_MAX_ROWS:
            return None
        has_dropout = self.__dict__.get("_has_dropout")
        if has_dropout is None:
            has_dropout = any(u.dropout_p > 0.0 for net in (self.encoder, self.decoder)
                              for u in net.plan().trunk + [u for h in net.plan().heads.values() for u in h])
            self.__dict__["_has_dropout"] = has_dropout
        if has_dropout:
            return None
        if nv.lib().btvae_prof_enabled():
            return None
        return (tuple(x.shape), tuple(c.shape), S, float(tk["beta"]), self.gemm_precision, id(optim), self._arena.data_ptr(),
                ck.get("max_norm") if clip else None, E.FOLD_BN_INTO_CONSUMER, E.BWD_SUMS_FROM_CONSUMERS,
                None if eps is None else tuple(eps.shape),
                tuple((off, cnt) for off, cnt, _ in optim._spans()))      # the captured optimiser launches: one span or two

    def _train_step(self, x, c, optim, eps=None):
        key = self._graph_key(x, c, optim, eps)
        out = None
        if key is not None:
            graphs = self.__dict__.setdefault("_step_graphs", {})
            seen = self.__dict__.setdefault("_step_seen", {})
            # a captured step has its products' kernels baked in: once the two-piece guard raises a range flag, what was
            # captured before no longer matches what the engine would launch (that site on three pieces)
            tripped = E.GUARD.tripped_sites() if self.gemm_precision == "fp32_split2" else 0
            if tripped != self.__dict__.get("_step_graph_tripped", 0):
                graphs.clear()
                seen.clear()
                self.__dict__["_step_graph_tripped"] = tripped
            g = graphs.get(key)
            if g is None and seen.get(key, 0) >= STEP_GRAPH_WARMUP and not self.__dict__.get("_step_graph_failed"):
                self._check_arena()
                optim.refresh_hyper(key[7])
                counters = ([getattr(net.plan(), "nbt_pending", 0) for net in (self.encoder, self.decoder)], optim._ensure().step_count)
                try:
                    g = _StepGraph(self, optim, self._prep(x, "inputs"), self._prep(c, "conditions"),
                                   None if eps is None else self._prep(eps, "eps"))
                except Exception as exc:          # leave the eager path in charge for the rest of the run
                    self.__dict__["_step_graph_failed"] = True
                    for net, v in zip((self.encoder, self.decoder), counters[0]):
                        net.plan().nbt_pending = v
                    optim._ensure().step_count = counters[1]
                    import warnings
                    warnings.warn(f"capturing the training step as a HIP graph failed ({exc}); continuing without it")
                    g = None
                else:
                    if len(graphs) >= 2:
                        graphs.clear()
                    graphs[key] = g
                    # capturing recorded the launches without running them, but the host-side counters of that pass
                    # (BatchNorm batches, optimiser steps) already moved: put them back, the replay below counts itself
                    for net, v in zip((self.encoder, self.decoder), counters[0]):
                        net.plan().nbt_pending = v
                    optim._ensure().step_count = counters[1]
            seen[key] = seen.get(key, 0) + 1
            if g is not None:
                optim.refresh_hyper(key[7])
                out = g.run(x, c, eps)
                for net in (self.encoder, self.decoder):
                    plan = net.plan()
                    plan.nbt_pending = getattr(plan, "nbt_pending", 0) + 1
                optim._ensure().step_count += 1
        if out is None:
            out = self._train_step_eager(x, c, optim, eps=eps)
        return {"loss": out[0], "elbo": out[1], "rll": out[2], "kl": out[3]}

    def train_core(self, inputs, optim, **_):
        return self._train_step(inputs, None, optim)

    def validate(self, valloader, num_mc_samples=1, device="cpu"):
        self.eval()
        tot = {"elbo": 0.0, "rll": 0.0, "kl": 0.0}
        with torch.no_grad():
            for inputs in valloader:
                x = self.move_to_device(inputs, device=device)
                fw = self._run_forward(x, None, num_mc_samples)
                out = self._loss_from_forward(fw, x, 1.0).tolist()
                for k, v in zip(("elbo", "rll", "kl"), out[1:]):
                    tot[k] += v * x.shape[0]
        n = len(valloader.dataset)
        return {k: v / n for k, v in tot.items()}

    def train_verbose(self, itx, loss_dict, pbar=None):
        msg = (f"Iteration: {itx} -- ELBO={loss_dict['elbo'].item():.4e} / RLL={loss_dict['rll'].item():.4e} / "
               f"KL={loss_dict['kl'].item():.4e}")
        (pbar.write if pbar is not None else print)(msg)
        E.GUARD.check()          # the read-outs above synchronised: what the two-piece guard recorded is final

    def move_to_device(self, inputs, device="cpu"):
        return inputs.to(device)

    def _prior_to(self, device):
        self.prior_params = {k: v.to(device) for k, v in self.prior_params.items()}

    def save(self, save_path=None, model_name="trained_model", device="cpu"):
        """``state_dict`` to ``<save_path>/<model_name>.pt`` from CPU copies, then back to ``device``
        (reference: src/vae_models.py:108-116)."""
        if save_path is None:
            save_path = self.log_dir
        was_training = self.training
        self.eval()
        self.to("cpu")
        self._prior_to("cpu")
        torch.save(self.state_dict(), os.path.join(save_path, model_name + ".pt"))
        self.to(device)
        self._prior_to(self.train_kwargs.get("device", device))
        # The reference leaves the model in eval mode here, so its fit() trains with frozen BatchNorm statistics
        # from a checkpoint until the next validation; that is treated as a defect and not reproduced.
        self.train(was_training)

    def load(self, load_path=None, model_name="trained_model"):
        if load_path is None:
            load_path = self.log_dir
        self.load_state_dict(torch.load(os.path.join(load_path, model_name + ".pt"), weights_only=True))
        self.to("cpu")
        self._prior_to("cpu")
        self.eval()

    def fit(self, trainloader, valloader=None, lr=1e-3, weight_decay=1e-4, gradient_clipping=False,
            gradient_clipping_kwargs={"max_norm": 1.0}, lr_scheduling=False,
            lr_scheduling_kwargs={"threshold": 0.1, "factor": 0.2, "patience": 5, "min_lr": 1e-5}, beta=1.0, num_mc_samples=1,
            epochs=1000, verbose_freq=100, tensorboard=True, tqdm_func=None, validation_freq=200, validation_mc_samples=1,
            device="cpu", earlystopping=False, earlystopping_kwargs={"patience": 5, "delta": 0.1}, save_epoch_freq=5,
            writer=None, **_):
        """Training loop with the reference's cadence (src/vae_models.py:126-238): validation every
        ``validation_freq`` iterations driving ReduceLROnPlateau(mode=max, threshold_mode=abs) and early
        stopping, console/scalar logging every ``verbose_freq``, checkpoint every ``save_epoch_freq`` epochs."""
        kw = dict(lr=lr, weight_decay=weight_decay, gradient_clipping=gradient_clipping,
                  gradient_clipping_kwargs=gradient_clipping_kwargs, lr_scheduling=lr_scheduling,
                  lr_scheduling_kwargs=lr_scheduling_kwargs, beta=beta, num_mc_samples=num_mc_samples, epochs=epochs,
                  verbose_freq=verbose_freq, tensorboard=tensorboard, validation_freq=validation_freq,
                  validation_mc_samples=validation_mc_samples, device=device, earlystopping=earlystopping,
                  earlystopping_kwargs=earlystopping_kwargs, save_epoch_freq=save_epoch_freq, _=_)
        self.train_kwargs.update(kw)

        def flat(d):
            out = {}
            for k, v in d.items():
                out.update(v) if k == "_" else out.__setitem__(k, v)
            return out

        if writer is None:
            writer = make_writer()
        self.log_dir = writer.log_dir
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "model_kwargs.json"), "w") as f:
            json.dump(flat(self.model_kwargs), f, indent=4)
        with open(os.path.join(self.log_dir, "train_kwargs.json"), "w") as f:
            json.dump(flat(self.train_kwargs), f, indent=4)
        if tensorboard and hasattr(writer, "file_writer"):
            ew = writer.file_writer.event_writer
            ew._logdir = os.path.join(ew._logdir, "tensorboard")

        pbar_e = pbar_i = None
        if tqdm_func is not None:
            per_epoch = len(trainloader.dataset) // trainloader.batch_sampler.batch_size + (trainloader.drop_last == False)  # noqa: E712
            pbar_e = tqdm_func(total=epochs, desc="Epoch", dynamic_ncols=True)
            pbar_i = tqdm_func(total=per_epoch, desc="Iteration in Epoch", dynamic_ncols=True)
        say = pbar_i.write if pbar_i is not None else print

        self.to(device)
        self._prior_to(device)
        optim = self.get_optimizer(lr=lr, weight_decay=weight_decay)
        scheduler = None
        if lr_scheduling:
            scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optim, mode="max", threshold_mode="abs", **lr_scheduling_kwargs)
        stopper = EarlyStopping(**earlystopping_kwargs) if earlystopping else None

        self.train()
        itx = 0
        # The interpreter's full collections walk every object the imports left behind (tens of milliseconds in which nothing
        # is launched: several steps' worth); once the loop's own objects exist they are moved out of the collector's
        # reach for the duration of the loop.  BTVAE_GC_FREEZE=0 leaves the collector alone.
        freeze_at = 8 if os.environ.get("BTVAE_GC_FREEZE", "1") != "0" else -1
        frozen = False
        try:
            for epx in range(1, epochs + 1):
                if pbar_e is not None:
                    pbar_e.update(1); pbar_i.reset()
                for batch in trainloader:
                    itx += 1
                    if itx == freeze_at:
                        gc.collect()
                        gc.freeze()
                        frozen = True
                    if pbar_i is not None:
                        pbar_i.update(1)
                    loss = self.train_core(self.move_to_device(batch, device=device), optim)
                    if valloader is not None and itx % validation_freq == 0:
                        val = self.validate(valloader, num_mc_samples=validation_mc_samples, device=device)
                        say(f"Validation -- ELBO={val['elbo']:.4e} / RLL={val['rll']:.4e} / KL={val['kl']:.4e}")
                        if scheduler is not None:
                            before = scheduler.get_last_lr()[0]
                            scheduler.step(val["elbo"])
                            after = scheduler.get_last_lr()[0]
                            if after != before:
                                say(f"Learning Rate Changed: {before} -> {after}")
                        if stopper is not None:
                            stopper(val["elbo"], pbar=pbar_i)
                        if tensorboard:
                            for k, tag in (("elbo", "Loss/ELBO"), ("rll", "Loss/RLL"), ("kl", "Loss/KL")):
                                writer.add_scalars(tag, {"val": val[k]}, itx)
                        self.train()
                    if itx % verbose_freq == 0:
                        self.train_verbose(itx, loss, pbar=pbar_i)
                        if tensorboard:
                            for k, tag in (("elbo", "Loss/ELBO"), ("rll", "Loss/RLL"), ("kl", "Loss/KL")):
                                writer.add_scalars(tag, {"train": loss[k]}, itx)
                if save_epoch_freq is not None and epx % save_epoch_freq == 0:
                    self.save(device=device)
                    (pbar_e.write if pbar_e is not None else print)(f"Model Saved at Epoch {epx} to {self.log_dir}")
                if stopper is not None and stopper.early_stop:
                    break
        finally:
            if frozen:
                gc.unfreeze()
        self.to("cpu")
        self._prior_to("cpu")
        self.eval()


class CVAE(VAE):
    """Conditional VAE: encoder sees x‖c, decoder sees z‖c (reference: src/vae_models.py:240-339)."""

    def __init__(self, input_dim=None, conditioner=None, latent_dim=10, distribution_dict={}, **_):
        # The reference first builds an unconditional encoder/decoder pair and then replaces it
        # (src/vae_models.py:251-256); doing the same keeps seeded initial weights identical.
        torch.nn.Module.__init__(self)
        self.model_kwargs = {"latent_dim": latent_dim, "distribution_dict": distribution_dict, "_": {"_": _}}
        self.latent_dim, self.distribution_dict = latent_dim, distribution_dict
        self.input_dim = input_dim
        throwaway_enc = get_distribution_model(input_dim=input_dim, output_dim=latent_dim, **distribution_dict["posterior"])
        throwaway_dec = get_distribution_model(input_dim=latent_dim, output_dim=input_dim, **distribution_dict["likelihood"])
        self.num_parameters = throwaway_enc._num_parameters() + throwaway_dec._num_parameters()
        del throwaway_enc, throwaway_dec
        self.condition_dim = conditioner.cond_dim
        self.encoder = get_distribution_model(input_dim=input_dim + self.condition_dim, output_dim=latent_dim, **distribution_dict["posterior"])
        self.decoder = get_distribution_model(input_dim=latent_dim + self.condition_dim, output_dim=input_dim, **distribution_dict["likelihood"])
        self.prior_params = get_prior_params(distribution_dict["posterior"]["dist_type"], latent_dim)
        self.train_kwargs = {}
        self.log_dir = None
        self._arena = None
        self._grad_arena = None
        self._dp_world = 1
        self._dp_active = False
        self._bn_sync = None
        self.gemm_precision = DEFAULT_GEMM_PRECISION
        self.set_gemm_precision(_.get("gemm_precision", DEFAULT_GEMM_PRECISION))
        self._finish_init()

    def forward(self, inputs, conditions, num_mc_samples=1):
        if conditions.ndim not in (2, 3):
            raise ValueError("Invalid shape for conditions. Must be 2D or 3D tensor.")
        if self._graph_mode():
            return self._forward_graph(inputs, conditions, num_mc_samples)
        fw = self._run_forward(inputs, conditions, num_mc_samples)
        return self._pack_outputs(fw)

    @torch.no_grad()
    def sample(self, condition, num_samples_prior=1, num_samples_likelihood=1):
        """Prior sampling (reference: src/vae_models.py:282-302): z ~ N(0,I) per (sample, row), decoder
        parameters ``[K,B,.]`` and ``num_samples_likelihood`` draws from each."""
        self._check_arena()
        K = num_samples_prior
        if condition.ndim == 2:
            three_d, B = False, condition.shape[0]
        elif condition.ndim == 3:
            if condition.shape[1] != K:
                raise ValueError("Condition samples must be equal to number of prior samples.")
            three_d, B = True, condition.shape[0]
        else:
            raise ValueError("Invalid shape for conditions. Must be 2D or 3D tensor.")
        cond = self._prep(condition, "condition")
        L = self.latent_dim
        z = torch.randn((K * B, L), device=cond.device)
        zeros, ones = torch.zeros(K * B if three_d else B, L, device=cond.device), None
        ones = torch.ones_like(zeros)
        if three_d:
            zc, _, _ = E.reparam_kl(zeros, ones, z, cond, K, True, B)
        else:
            zc, _, _ = E.reparam_kl(zeros, ones, z, cond, K, False, B)
        lik, _ = E.net_forward(self.decoder.plan(), zc, self.training, None, self.decoder.head_specs())
        param_dict = {k: v.view(K, B, -1) for k, v in lik.items()}
        samples = self.decoder.sample(param_dict, num_samples=num_samples_likelihood)
        return {"params": param_dict, "samples": samples}

    @torch.no_grad()
    def reconstruct(self, inputs, conditions, num_mc_samples=1):
        return self.forward(inputs=inputs, conditions=conditions, num_mc_samples=num_mc_samples)

    @torch.no_grad()
    def loglikelihood(self, inputs, conditions, num_mc_samples=1):
        """Importance-weighted log-likelihood per profile, -log K + logsumexp_k(log p(x|z_k,c) + log p(z_k) - log q(z_k|x,c)).
        The reference's method of this name cannot run (``torch.logsumexp`` without ``dim``, src/vae_models.py:315); this
        returns what its formula describes, the quantity ``testing_lib.mass_loglikelihood`` computes (src/testing_lib.py:14)."""
        from .testing_lib import iw_loglikelihood_device
        ll, _, _ = iw_loglikelihood_device(self, inputs, conditions, num_mc_samples, min(int(num_mc_samples), 100))
        return ll

    def train_core(self, inputs, optim, **_):
        x, c = inputs
        return self._train_step(x, c, optim)

    @torch.no_grad()
    def validate(self, valloader, num_mc_samples=1, beta=1.0, device="cpu"):
        self.eval()
        tot = {"elbo": 0.0, "rll": 0.0, "kl": 0.0}
        for batch in valloader:
            x, c = self.move_to_device(batch, device=device)
            fw = self._run_forward(x, c, num_mc_samples)
            out = self._loss_from_forward(fw, self._prep(x, "inputs"), beta).tolist()
            for k, v in zip(("elbo", "rll", "kl"), out[1:]):
                tot[k] += v * x.shape[0]
        n = len(valloader.dataset)
        return {k: v / n for k, v in tot.items()}

    def move_to_device(self, inputs, device="cpu"):
        return [t.to(device) for t in inputs]
