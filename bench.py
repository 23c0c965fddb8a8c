#!/usr/bin/env python3
"""Benchmark of the config0 ELBO training step on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1, one rank per GPU)

A step = one ``CVAE.train_core`` (forward, ELBO, backward, global-norm clip, Adam) on one resident batch of
synthetic daily load profiles of the config0 shape (D=48, C=4, L=48, H=1000, V=100, S=16 MC samples,
batch 4096 per GPU, fp32).  Inputs are generated on the device before the timed region.  Rank 0 prints one
JSON line; ``value`` is whole-job samples/s.  Also reported: the roofline of the dominant kernel (the decoder's hidden
Linear product, in the default mode fp32 operands as two scaled fp16 pieces on the fp16 matrix pipe) from HIP-event
timing on the launch stream inside the timed region; the same step at the other operand precisions; the K=1000
importance-weighted scoring leg (``secondary``) and the pseudo-Gibbs imputation leg (``imputation``), each with the
roofline of its own dominant kernel; an end-to-end run of ``scripts/training.main(config0)`` (``e2e``); the
wide-conditioning bf16 variant (``wide_conditioning``); and CPU baselines (the oracle restatement timed on the host
cores, rank 0, N=1 only).  ``--scaling strong`` keeps the global batch at ``--batch`` and splits it over the ranks.
"""
import argparse
import contextlib
import gc
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# effective config0 (reference: config_files/config0.json) at the benchmark's batch of 4096 per GPU
CONFIG0_MODEL = {
    "latent_dim": 48,
    "distribution_dict": {
        "posterior": {"dist_type": "normal", "learn_sigma": True, "sigma_lim": 0.05, "total_max_std": 3.0, "mu_upper_lim": 5.0,
                      "mu_lower_lim": -5.0, "num_neurons": 1000, "num_hidden_layers": 2, "dropout": False, "dropout_rate": 0.2,
                      "batch_normalization": True, "resnet": False},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "learn_sigma": True, "sigma_lim": 1e-4, "marginal_std_lim": 0.1,
                       "total_max_std": 6.0, "mu_upper_lim": 5.0, "mu_lower_lim": -3, "num_neurons": 1000, "num_hidden_layers": 2,
                       "dropout": False, "dropout_rate": 0.2, "batch_normalization": True, "resnet": False}}}
CONFIG0_TRAIN = {"lr": 1e-3, "beta": 1.0, "num_mc_samples": 16, "weight_decay": 1e-5, "gradient_clipping": True,
                 "gradient_clipping_kwargs": {"max_norm": 1.0}}
D, C, SEED = 48, 4, 2112
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X fp32 matrix peak (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD)
BREAKDOWN_STEPS = 3             # untimed steps with every launch event-timed, for the per-family table
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X bf16 dense matrix peak (v_mfma_f32_32x32x16_bf16, 1024 FLOP/clk/SIMD)
SPLIT_TERMS = 6                  # bf16 MFMAs issued per fp32 product in the split mode


def kernel_sources_hash() -> str:
    """sha256 over the kernel sources: what a recorded PMC figure was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "behavio_temporal_vae_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def recorded_traffic(leg: str, precision: str, M: int):
    """Beyond-L2 bytes per launch of a leg's dominant kernel from the PMC passes recorded in profiles/traffic.json
    (``tools/profiling/traffic_json.py``: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over this script, FETCH_SIZE
    doubled as gfx950 needs), together with where they come from.  The figures are not collected live, so they are only
    reported when the kernel sources still hash to what they were measured on; otherwise ``traffic`` is null and the
    note says so."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, "no profiles/traffic.json"
    entry = rec.get("legs", {}).get(f"{leg}:{precision}:{M}")
    if entry is None:
        return None, f"profiles/traffic.json holds no pass for {leg} / {precision} / M={M}"
    src = f"profiles/traffic.json ({rec.get('command', '?')}; commit {rec.get('commit', '?')}): FETCH_SIZE x2 {entry['fetch_x2_bytes']:.0f} + WRITE_SIZE {entry['write_bytes']:.0f} bytes, median per dispatch"
    if rec.get("kernel_sources_hash") != kernel_sources_hash():
        return None, "stale: kernel sources changed since " + src
    return entry["fetch_x2_bytes"] + entry["write_bytes"], src


KERNEL_NAME = {"fp32_split": "gemm_split_mfma", "fp32_split2": "gemm_split_mfma", "fp32": "gemm_f32_mfma", "bf16": "gemm_bf16_mfma"}
PRECISION_NOTE = {
    "fp32_split": "Linear products: fp32 operands as exact 3-way bf16 splits, 6 bf16 MFMAs per product with fp32 accumulation; "
                  "error against fp64 at or below the fp32 MFMA instruction's (tools/gemm_accuracy.py)",
    "fp32_split2": "Linear products: fp32 operands as two fp16 pieces of the per-tensor-scaled value, 3 fp16 MFMAs per product with fp32 "
                   "accumulation (products without tracked operand maxima: the 3-way bf16 split)",
    "fp32": "Linear products on the fp32 MFMA instruction",
    "bf16": "Linear products on bf16-rounded operands",
}


class Cdim:
    def __init__(self, c):
        self.cond_dim = c


def synth_batch(rng, B):
    """Synthetic profiles and calendar conditions of the config0 shape: x ~ clamp(N(1,1),-3,5) with 10 % of
    entries at the zero id -3; months U{1..12}, weekdays U{0..6} circularly encoded as (cos.., sin..)."""
    x = np.clip(rng.normal(1.0, 1.0, size=(B, D)), -3, 5)
    x[rng.random((B, D)) < 0.1] = -3.0
    months = rng.integers(1, 13, size=(B, 1)); weekdays = rng.integers(0, 7, size=(B, 1))
    am = (months - 1) / 12 * 2 * np.pi; aw = weekdays / 7 * 2 * np.pi
    c = np.concatenate([np.cos(am), np.sin(am), np.cos(aw), np.sin(aw)], axis=1)
    return x.astype(np.float32), c.astype(np.float32)


def flops_per_sample(S):
    """Algorithmic FLOPs of one training sample: encoder 5,148,000 MAC + S x decoder 5,200,000 MAC forward,
    x3 for forward + dgrad + wgrad, x2 FLOP/MAC (dictionary-Gaussian covariance GEMMs not counted)."""
    return (5_148_000 + S * 5_200_000) * 2 * 3


def _cpu_info():
    model, phys = "unknown", None
    try:
        cores = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                cores.add((pid, cid))
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys


def _cpu_threads():
    cores = os.cpu_count() or 1
    threads = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(threads)
    return threads


def cpu_baseline(S=16):
    """The CPU oracle's train step (same algorithm as the reference's CPU path) on the host cores: B=1024 (the batch
    config0.json ships) and B=4096 (the benchmark's), plus the B=1024 step under ``torch.autograd.set_detect_anomaly``
    -- the reference turns that on at import (src/vae_models.py:10), which adds a NaN scan to every backward op."""
    from oracle import refimpl as R
    threads = _cpu_threads()
    model, phys = _cpu_info()
    spec = R.model_spec(D, C, CONFIG0_MODEL)
    rng = np.random.default_rng(SEED)
    g = torch.Generator().manual_seed(SEED)

    def run(B, steps, anomaly=False):
        sd = R.init_state(spec, SEED)
        st = R.AdamState()
        ts = []
        with torch.autograd.set_detect_anomaly(anomaly):
            for i in range(steps + 1):
                x, c = synth_batch(rng, B)
                eps = torch.randn(S, B, spec.latent_dim, generator=g)
                t0 = time.perf_counter()
                R.train_step(sd, spec, st, torch.from_numpy(x), torch.from_numpy(c), eps, lr=1e-3, weight_decay=1e-5, beta=1.0, max_norm=1.0)
                ts.append(time.perf_counter() - t0)
        return sum(ts[1:]) / steps

    t1024 = run(1024, 2)
    t1024_anom = run(1024, 1, anomaly=True)
    t4096 = run(4096, 1)
    return {"value": 4096 / t4096, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"1 train step of B=4096, S={S} (config0 model) after 1 warm-up step, {t4096:.2f} s/step",
            "cpu_model": model, "physical_cores": phys,
            "variants": {"B=1024": {"value": 1024 / t1024, "s_per_step": t1024, "steps": 2},
                         "B=1024, anomaly detection on (as the reference ships)": {"value": 1024 / t1024_anom, "s_per_step": t1024_anom, "steps": 1},
                         "B=4096": {"value": 4096 / t4096, "s_per_step": t4096, "steps": 1}}}


GIPUZKOA_MODEL = {
    "latent_dim": 24,
    "distribution_dict": {
        "posterior": {"dist_type": "normal", "learn_sigma": True, "sigma_lim": 0.05, "average_max_std": 1.0, "mu_upper_lim": 5.0,
                      "mu_lower_lim": -5.0, "num_neurons": 1000, "num_hidden_layers": 2, "dropout": True, "dropout_rate": 0.2,
                      "batch_normalization": True, "resnet": False},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "learn_sigma": True, "sigma_lim": 1e-6, "marginal_std_lim": 0.05,
                       "average_max_std": 1.0, "mu_upper_lim": 5.0, "mu_lower_lim": -3, "num_neurons": 1000, "num_hidden_layers": 2,
                       "dropout": True, "dropout_rate": 0.2, "batch_normalization": True, "resnet": False}}}


def cpu_scoring_baseline(K=1000, N=32):
    """The CPU oracle's K-sample importance-weighted scoring on the host cores (gipuzkoa-shaped model, eval mode):
    one warm-up call on N/4 profiles, then one timed call on N."""
    from oracle import refimpl as R
    threads = _cpu_threads()
    spec = R.model_spec(24, C, GIPUZKOA_MODEL)
    sd = R.init_state(spec, SEED)
    g = torch.Generator().manual_seed(SEED)
    x = torch.randn(N, 24, generator=g).clamp(-3, 5)
    ang = torch.rand(N, 2, generator=g) * (2 * math.pi)
    c = torch.cat([torch.cos(ang), torch.sin(ang)], 1)
    with torch.no_grad():
        R.iw_loglikelihood(sd, spec, x[:N // 4], c[:N // 4], torch.randn(K, N // 4, spec.latent_dim, generator=g))
        eps = torch.randn(K, N, spec.latent_dim, generator=g)
        t0 = time.perf_counter()
        R.iw_loglikelihood(sd, spec, x, c, eps)
        dt = time.perf_counter() - t0
    return {"value": N / dt, "unit": "profiles/s", "cores": threads, "kind": "port",
            "sample": f"{N} profiles at K={K} in one call after a warm-up call on {N // 4}, {dt:.2f} s"}


def guard_report():
    """What the guard of the two-piece products has seen in this process so far: launches routed to the three-piece
    kernel because an operand had a region far below its maximum, sites that raised that flag, and -- raising -- any
    product whose accumulators overflowed.  Taken once right after the headline's timed steps (must show no fallback)
    and once at the end over every leg (`all_legs`: the end-to-end training leg's decoder sigma-head chain trips it)."""
    from behavio_temporal_vae_amd import engine as E
    E.GUARD.check()
    return {"fallbacks": int(E.GUARD.fallbacks), "sites_flagged": E.GUARD.tripped_sites(), "nonfinite": 0,
            "note": "fallbacks must be 0 on the headline run: every Linear product of the timed steps ran on two fp16 pieces"}


def _peak(precision):
    if precision == "fp32_split":
        return PEAK_BF16_MFMA_TFLOPS / SPLIT_TERMS, "fp32-equivalent FLOP/s: bf16 dense peak 2500 / 6 bf16 MFMAs per fp32 product"
    if precision == "fp32_split2":
        return PEAK_BF16_MFMA_TFLOPS / 3, "fp32-equivalent FLOP/s: fp16 dense peak 2500 / 3 fp16 MFMAs per fp32 product"
    return (PEAK_F32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS), "dense matrix peak of the operand type"


def _dominant_roofline(nv, precision, what, M, algorithmic_bytes, leg):
    """Roofline block of the launches of the gemm_fwd family with the largest FLOP count, from the HIP events recorded
    since ``prof_enable`` (call after a device sync)."""
    dom_ms, dom_fl, dom_n = nv.prof_dominant("gemm_fwd")
    nv.prof_collect()
    nv.prof_enable(False)
    peak, note = _peak(precision)
    achieved = dom_fl * dom_n / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    return {"bound": "mfma", "kernel": what, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "peak_note": note, "flops_per_launch": dom_fl, "avg_launch_ms": dom_ms / max(dom_n, 1), "launches": dom_n,
            "traffic": recorded_traffic(leg, precision, M)[0], "traffic_source": recorded_traffic(leg, precision, M)[1],
            "algorithmic_bytes": algorithmic_bytes}


def scoring_benchmark(dev, rank, world, n_profiles, K, chunk):
    """K-sample importance-weighted log-likelihood of the config0_gipuzkoa-shaped model (D=24, L=24, C=4, H=1000,
    V=100, eval mode): every rank scores its own ``n_profiles`` resident profiles (weak scaling), one reduce of
    (sum ll, count) at the end.  Returns whole-job profiles/s and the roofline of the leg's dominant kernel (the
    decoder's hidden Linear in eval mode, BatchNorm folded into the epilogue, at M = chunk x profiles rows)."""
    import torch.distributed as dist
    from behavio_temporal_vae_amd import _native as nv
    from behavio_temporal_vae_amd import testing_lib
    from behavio_temporal_vae_amd.parallel import reduce_loglik
    from behavio_temporal_vae_amd.vae_models import CVAE
    torch.manual_seed(SEED)
    with contextlib.redirect_stdout(sys.stderr):
        model = CVAE(input_dim=24, conditioner=Cdim(C), **GIPUZKOA_MODEL)
    model = model.to(dev).eval()
    rng = np.random.default_rng(SEED + 100 + rank)
    x = np.clip(rng.normal(1.0, 1.0, size=(n_profiles, 24)), -3, 5).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, size=(n_profiles, 2))
    c = np.concatenate([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    xd, cd = torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)
    testing_lib.iw_loglikelihood_device(model, xd[:256], cd[:256], 2 * chunk, chunk)     # warm-up
    testing_lib.iw_loglikelihood_device(model, xd, cd, chunk, chunk)                     # ... and once at the full row count
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    M = chunk * n_profiles
    nv.prof_filter(["gemm_fwd"], 0.9 * 2.0 * M * 1000 * 1000)
    nv.prof_enable(True)
    t0 = time.perf_counter()
    ll, _, _ = testing_lib.iw_loglikelihood_device(model, xd, cd, K, chunk)
    tot, cnt = reduce_loglik(float(ll.double().sum().item()), n_profiles, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from behavio_temporal_vae_amd import engine as E
    pairs = E._eval_pairs_ok(M, 1000) and model.gemm_precision == "fp32_split2"
    kname = ("gemm_pq256<relu+affine, 256 x 256 tile, two fp16 pieces, activations in and (trunk) out in the pair format, operands by global->LDS DMA>"
             if pairs else f"{KERNEL_NAME[model.gemm_precision]}<k-contig,k-contig,relu+affine(+residual),pre-split weight>")
    roof = _dominant_roofline(nv, model.gemm_precision, f"{kname} M={M} N=K=1000 (decoder hidden Linear, eval mode)", M,
                              4.0 * (2.0 * M * 1000 + 1000 * 1000), "scoring_fwd_pairs" if pairs else "scoring_fwd")
    roof["activations_in_pair_format"] = bool(pairs)
    nv.prof_filter(None, 0.0)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    flops = (5_076_000 + K * 5_152_000) * 2.0 * cnt
    return {"metric": "MC-scored profiles/sec at K=1000", "value": cnt / dt, "unit": "profiles/s", "K": K, "profiles": cnt,
            "seconds": dt, "mean_loglik": tot / cnt, "model_tflops": flops / dt / 1e12, "roofline": roof,
            "config": {"workload": "config0_gipuzkoa-shaped IW log-likelihood, eval mode", "gemm_precision": model.gemm_precision,
                       "input_dim": 24, "latent_dim": 24,
                       "hidden": 1000, "vocab": 100, "mc_chunk": chunk, "per_gpu_profiles": n_profiles}}


class _ReadyConditions:
    """Conditioner stand-in that hands a ready [B, C] array through (a deterministic conditioner's tags / types)."""
    tags, types, transformers, cond_dim = ["months"], ["circ"], {}, C

    def transform(self, cs, num_samples=1):
        return cs["c"]


def imputation_benchmark(dev, rows, sweeps):
    """Pseudo-Gibbs imputation sweeps (src/imputation_lib.py:59-98) at the reference's default batch and window
    (scripts/collect_imputation_results.py:160-164: 25,000 profiles, the last 24 of config0's 48 coordinates missing):
    one sweep = eval-mode forward with one latent draw, decoder covariance, conditional Gaussian per profile, one draw,
    masked write-back.  Dominant kernel: the hidden Linear in eval mode at M = rows."""
    from behavio_temporal_vae_amd import _native as nv
    from behavio_temporal_vae_amd import imputation_lib as IL
    from behavio_temporal_vae_amd.vae_models import CVAE
    torch.manual_seed(SEED)
    with contextlib.redirect_stdout(sys.stderr):
        model = CVAE(input_dim=D, conditioner=Cdim(C), **CONFIG0_MODEL).to(dev).eval()
    rng = np.random.default_rng(SEED + 7)
    x, c = synth_batch(rng, rows)
    x = torch.from_numpy(x).to(dev)
    mask = torch.zeros(rows, D, dtype=torch.bool, device=dev)
    mask[:, D - 24:] = True
    x[mask] = 1.0
    cs = {"c": c}
    with contextlib.redirect_stdout(sys.stderr):
        IL.pseudo_gibbs(model, x, mask, _ReadyConditions(), cs, num_samples=10, num_iter=3, device=dev)
        torch.cuda.synchronize()
        nv.prof_filter(["gemm_fwd"], 0.9 * 2.0 * rows * 1000 * 1000)
        nv.prof_enable(True)
        t0 = time.perf_counter()
        IL.pseudo_gibbs(model, x, mask, _ReadyConditions(), cs, num_samples=10, num_iter=sweeps, device=dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    roof = _dominant_roofline(nv, model.gemm_precision,
                              f"{KERNEL_NAME[model.gemm_precision]}<k-contig,k-contig,relu+affine(+residual),pre-split weight> M={rows} N=K=1000 "
                              "(hidden Linear, eval mode)", rows, 4.0 * (2.0 * rows * 1000 + 1000 * 1000), "imputation_fwd")
    nv.prof_filter(None, 0.0)
    flops = (5_148_000 + 5_200_000) * 2.0 * rows * sweeps
    return {"metric": "pseudo-Gibbs imputation profile-sweeps/sec", "value": rows * sweeps / dt, "unit": "profile-sweeps/s",
            "sweeps_per_s": sweeps / dt, "ms_per_sweep": dt / sweeps * 1e3, "rows": rows, "sweeps": sweeps,
            "model_tflops": flops / dt / 1e12, "roofline": roof,
            "config": {"workload": "config0-shaped pseudo-Gibbs sweep, 24 of 48 coordinates missing, eval mode",
                       "gemm_precision": model.gemm_precision}}


def cpu_imputation_baseline(rows=512, sweeps=2):
    """The CPU oracle's pseudo-Gibbs sweep on the host cores (config0 shape, 24 of 48 coordinates missing)."""
    from oracle import refimpl as R
    threads = _cpu_threads()
    spec = R.model_spec(D, C, CONFIG0_MODEL)
    sd = R.init_state(spec, SEED)
    rng = np.random.default_rng(SEED + 7)
    x, c = synth_batch(rng, rows)
    x, c = torch.from_numpy(x), torch.from_numpy(c)
    mask = torch.zeros(rows, D, dtype=torch.bool)
    mask[:, D - 24:] = True
    x[mask] = 1.0
    g = torch.Generator().manual_seed(SEED)
    ez, ex = torch.randn(sweeps + 1, rows, spec.latent_dim, generator=g), torch.randn(sweeps + 1, rows, 24, generator=g)
    with torch.no_grad():
        R.pseudo_gibbs(sd, spec, x, mask, c, ez[:1], ex[:1], 10)
        t0 = time.perf_counter()
        R.pseudo_gibbs(sd, spec, x, mask, c, ez[1:], ex[1:], 10)
        dt = time.perf_counter() - t0
    return {"value": rows * sweeps / dt, "unit": "profile-sweeps/s", "cores": threads, "kind": "port",
            "sample": f"{sweeps} sweeps over {rows} profiles after a warm-up sweep, {dt:.2f} s"}


def e2e_benchmark(dev, iterations):
    """``scripts/training.main(config0.json)`` end to end on one GPU: synthetic set of ``iterations`` x 1024 profiles
    (one epoch at the batch the config ships), the device batch loader, validation and console logging at the
    configured cadence (every 1000 / 250 iterations), checkpoint at the end.  Timed around ``train_model`` (loaders +
    ``fit``)."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("btvae_training_entry", os.path.join(ROOT, "scripts", "training.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    cfg = tr.load_config(os.path.join(ROOT, "config_files", "config0.json"))
    B = cfg["train"]["batch_size"]
    cfg["train"].update({"epochs": 1, "device": str(dev), "save_epoch_freq": None})
    cfg["save_dir"] = tempfile.mkdtemp(prefix="btvae_e2e_")
    timing = {}
    inner = tr.train_model

    def timed(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = inner(*a, **k)
        torch.cuda.synchronize()
        timing["s"] = time.perf_counter() - t0
        return out
    tr.train_model = timed
    torch.manual_seed(SEED)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):
        tr.main(cfg, n_train=iterations * B, n_val=8192)
    total = time.perf_counter() - t0
    return {"metric": "ELBO train samples/sec, scripts/training.main(config0.json) end to end", "value": iterations * B / timing["s"],
            "unit": "samples/s", "iterations": iterations, "batch": B, "train_model_seconds": timing["s"], "main_seconds": total,
            "includes": "device batch loader (shuffled index stream, on-device gather), validation every "
                        f"{cfg['train']['validation_freq']} iterations over 8192 profiles at S={cfg['train']['validation_mc_samples']}, "
                        f"console logging every {cfg['train']['verbose_freq']} iterations, LR scheduling / early-stopping bookkeeping"}


WIDE_MODEL = {
    "latent_dim": 24, "gemm_precision": "bf16",
    "distribution_dict": {
        "posterior": {"dist_type": "normal", "learn_sigma": True, "sigma_lim": 0.05, "mu_upper_lim": 5.0, "mu_lower_lim": -5.0,
                      "num_neurons": 1000, "num_hidden_layers": 2, "dropout": False, "batch_normalization": True},
        "likelihood": {"dist_type": "dict-gauss", "vocab_size": 100, "learn_sigma": True, "sigma_lim": 1e-4, "marginal_std_lim": 0.1,
                       "mu_upper_lim": 5.0, "mu_lower_lim": -3, "num_neurons": 1000, "num_hidden_layers": 2, "dropout": False,
                       "batch_normalization": True}}}


def wide_benchmark(dev, B, S, steps):
    """Sweep-style wide conditioning (scripts/config_sweep.py:76-81,89): months + weekdays + day-before profile + T=500 user
    topics -> C = 528, D = L = 24, H = 1000, bf16 GEMM operands.  A train step draws its batch's Dirichlet user
    conditions on the device (one draw per profile, 2-D conditions, as training feeds them) and runs ``train_core``; a
    validation-style step draws S sampled conditions per profile ([B, S, C]) and runs the eval forward + ELBO."""
    from behavio_temporal_vae_amd.conditioning_lib import Conditioner
    from behavio_temporal_vae_amd.datasets import DeviceConditionEncoder
    from behavio_temporal_vae_amd.vae_models import CVAE
    rng = np.random.default_rng(SEED + 5)
    n, T = 4 * B, 500
    cset = {"months": rng.integers(1, 13, size=(n, 1)), "weekdays": rng.integers(0, 7, size=(n, 1)),
            "day_befores": np.clip(rng.normal(1.0, 1.0, size=(n, 24)), -3, 5), "users": rng.gamma(0.05, 1.0, size=(n, T)) + 0.01}
    cond = Conditioner(tags=["months", "weekdays", "day_befores", "users"], supports=[list(range(1, 13)), list(range(7)), [-3, 5], [0.01, 10.0]],
                       types=["circ", "circ", "identity", "dir"], condition_set=cset)
    torch.manual_seed(SEED)
    with contextlib.redirect_stdout(sys.stderr):
        model = CVAE(input_dim=24, conditioner=cond, **WIDE_MODEL).to(dev).train()
    model.train_kwargs = dict(CONFIG0_TRAIN, num_mc_samples=S)
    optim = model.get_optimizer(lr=1e-3, weight_decay=1e-5)
    enc = DeviceConditionEncoder(cond, cset, dev)
    x = torch.from_numpy(np.clip(rng.normal(1.0, 1.0, size=(n, 24)), -3, 5).astype(np.float32)).to(dev)
    idx = [torch.arange(i * B, (i + 1) * B, device=dev) for i in range(4)]

    def train_step(i):
        j = idx[i % 4]
        return model.train_core((x[j], enc.encode(j, 1)[:, 0].contiguous()), optim)

    def val_step(i):
        j = idx[i % 4]
        with torch.no_grad():
            fw = model._run_forward(x[j], enc.encode(j, S), S)
            return model._loss_from_forward(fw, x[j], 1.0)

    out = {}
    for name, fn in (("train", train_step), ("validation", val_step)):
        model.train() if name == "train" else model.eval()
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            r = fn(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        last = float(r["loss"].item()) if isinstance(r, dict) else float(r[0].item())
        out[name] = {"value": B * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "last_loss": last}
    mac = (552_000 + 5_000_000 + 2 * 24_000) + S * (552_000 + 5_000_000 + 24_000 + 100_000)
    out["train"]["model_tflops"] = mac * 2 * 3 * out["train"]["value"] / 1e12
    # roofline of the forward products of the train step: the hidden layers (M = S B, N = K = 1000) and, separately, the
    # decoder's first layer with its K = L + C = 552 reduction -- HIP events around those launches in extra steps
    from behavio_temporal_vae_amd import _native as nv
    model.train()
    hidden_fl, first_fl = 2.0 * S * B * 1000 * 1000, 2.0 * S * B * 1000 * 552
    roof = {}
    for tag, lo in (("hidden", 0.9 * hidden_fl), ("hidden_and_first", 0.9 * first_fl)):
        nv.prof_filter(["gemm_fwd"], lo)
        nv.prof_enable(True)
        for i in range(3):
            train_step(i)
        torch.cuda.synchronize()
        roof[tag] = nv.prof_collect()["gemm_fwd"]
        nv.prof_enable(False)
    nv.prof_filter(None, 0.0)
    h_ms, h_fl, _, h_n = roof["hidden"]
    f_ms, f_fl, f_n = roof["hidden_and_first"][0] - h_ms, roof["hidden_and_first"][1] - h_fl, roof["hidden_and_first"][3] - h_n
    blocks = {}
    for tag, ms, fl, cnt, shape in (("hidden_layer", h_ms, h_fl, h_n, f"M={S * B} N=K=1000"), ("first_layer_k552", f_ms, f_fl, f_n, f"M={S * B} N=1000 K=552")):
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        kname = ("gemm_pq256<relu+stats, 256 x 256 tile, bf16 in / bf16 out, operands by global->LDS DMA>" if tag == "hidden_layer"
                 else "gemm_bf16_mfma<k-contig,k-contig,relu+stats, fp32 in / bf16 out>")
        blocks[tag] = {"bound": "mfma", "kernel": f"{kname} {shape}", "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS,
                       "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS, "avg_launch_ms": ms / max(cnt, 1), "launches": cnt,
                       "traffic": recorded_traffic("wide_" + tag, "bf16", S * B)[0], "traffic_source": recorded_traffic("wide_" + tag, "bf16", S * B)[1]}
    out["roofline"] = blocks
    return {"metric": "ELBO train samples/sec, wide conditioning (C=528), bf16 GEMM operands", **out,
            "config": {"workload": "config_sweep-style: C=528 (2+2+24+500), D=L=24, H=1000, V=100, on-device Dirichlet conditions",
                       "gemm_precision": "bf16", "per_gpu_batch": B, "num_mc_samples": S}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--option-steps", "--bf16-steps", dest="option_steps", type=int, default=10,
                    help="extra train steps per alternative GEMM precision (native fp32 MFMA, bf16 operands), reported beside the headline (0 = skip)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch per GPU; strong: --batch is the global batch, split over the GPUs")
    ap.add_argument("--scoring-profiles", type=int, default=32768, help="per-GPU profiles of the K=1000 scoring leg (0 = skip)")
    ap.add_argument("--imputation-rows", type=int, default=25000, help="profiles of the pseudo-Gibbs leg (0 = skip; rank 0, N=1 only)")
    ap.add_argument("--imputation-sweeps", type=int, default=30)
    ap.add_argument("--e2e-iterations", type=int, default=1000, help="iterations of the scripts/training.main(config0) leg (0 = skip; N=1 only)")
    ap.add_argument("--wide-steps", type=int, default=10, help="steps of the wide-conditioning bf16 leg (0 = skip; N=1 only)")
    ap.add_argument("--scoring-k", type=int, default=1000)
    ap.add_argument("--scoring-chunk", type=int, default=25, help="latent draws per decoder pass")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    # Rehearsal on a one-GPU box: BTVAE_BENCH_BACKEND=gloo with BTVAE_BENCH_ONE_GPU=1 runs every rank on device 0
    # (RCCL refuses two ranks on one device); it exercises the multi-rank control flow, not the collectives' speed.
    if os.environ.get("BTVAE_BENCH_ONE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    backend = os.environ.get("BTVAE_BENCH_BACKEND", "nccl")
    n_ranks_seen = 1
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=10))
        # what the collectives themselves see: a sum of ones over the backend the gradients will use
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        n_ranks_seen = int(round(float(ones.item())))
        if n_ranks_seen != world or dist.get_world_size() != world:
            raise SystemExit(f"rank {rank}: the process group spans {n_ranks_seen} ranks, expected {world}")

    from behavio_temporal_vae_amd import _native as nv
    from behavio_temporal_vae_amd.vae_models import CVAE

    torch.manual_seed(SEED)
    with contextlib.redirect_stdout(sys.stderr):   # the constructors print like the reference's; keep stdout for the JSON line
        model = CVAE(input_dim=D, conditioner=Cdim(C), **CONFIG0_MODEL)
    model = model.to(dev).train()
    model.train_kwargs = dict(CONFIG0_TRAIN)
    if world > 1:
        model.enable_data_parallel()
    optim = model.get_optimizer(lr=CONFIG0_TRAIN["lr"], weight_decay=CONFIG0_TRAIN["weight_decay"])
    if args.scaling == "strong" and args.batch % world:
        raise SystemExit(f"--scaling strong needs --batch ({args.batch}) divisible by the number of GPUs ({world})")
    B, S = (args.batch // world if args.scaling == "strong" else args.batch), CONFIG0_TRAIN["num_mc_samples"]
    rng = np.random.default_rng(SEED + rank)
    pool = []
    for _ in range(4):
        x, c = synth_batch(rng, B)
        pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)))
    torch.manual_seed(SEED + rank)

    def step(i):
        x, c = pool[i % len(pool)]
        return model.train_core((x, c), optim)

    for i in range(args.warmup):
        loss = step(i)
    # A full (generation-2) collection of the interpreter's garbage collector walks every object torch's import left
    # behind: ~70 ms of host time during which nothing is launched, 5 steps' worth.  What exists now is moved out of the
    # collector's reach, as long-running training loops do; the steps' own garbage is still collected.
    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Inside the timed region only the dominant kernel is bracketed by HIP events (the decoder's hidden Linear
    # forward, 3 launches a step): two event records per launch cost stream time, 0.7 ms a step when every launch of
    # the step carries them.  The per-family breakdown comes from a few extra, untimed steps afterwards.
    # At launch-bound batch sizes the step is replayed from a HIP graph, inside which single launches cannot be bracketed
    # by events: there the dominant kernel is timed in untimed eager steps right after the timed region instead.
    dominant_flops = 2.0 * (S * B) * 1000 * 1000
    from behavio_temporal_vae_amd import vae_models as VM
    graph_mode = os.environ.get("BTVAE_STEP_GRAPH", "auto")
    graphed = world == 1 and graph_mode != "0" and (S * B <= VM.STEP_GRAPH_MAX_ROWS or graph_mode == "1")
    if not graphed:
        nv.prof_filter(["gemm_fwd"], 0.9 * dominant_flops)
        nv.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last_timed = loss
    if graphed:
        nv.prof_filter(["gemm_fwd"], 0.9 * dominant_flops)
        nv.prof_enable(True)
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
    dom_ms, dom_fl, dom_n = nv.prof_dominant("gemm_fwd")
    # that shape runs on two kernels in the two-piece mode: the 256-wide DMA-fed one where the input is a materialised
    # BatchNorm output (kept in the pair format), the 128-wide one behind a folded BatchNorm.  The one with more launches
    # is the dominant kernel; the other is reported beside it.
    by_kernel = {k: nv.prof_dominant("gemm_fwd", k) for k in ("pairs", "plain")}
    dom_kernel = "pairs" if by_kernel["pairs"][2] > by_kernel["plain"][2] else "plain"
    if by_kernel[dom_kernel][1] >= 0.999 * dom_fl:
        dom_ms, dom_fl, dom_n = by_kernel[dom_kernel]
    headline_guard = guard_report()       # of the warm-up and the timed steps: before any other leg runs
    nv.prof_collect()
    nv.prof_filter(None, 0.0)
    nv.prof_enable(True)
    # the other two product families at the same shape (every hidden-layer dX product carries a BatchNorm backward or its
    # reduction in its epilogue; the weight gradients are split reductions), from the same untimed steps
    family_dominant = {}
    for i in range(BREAKDOWN_STEPS):
        step(i)
    torch.cuda.synchronize()
    for fam in ("gemm_dgrad", "gemm_wgrad"):
        f_ms, f_fl, f_n = nv.prof_dominant(fam)
        family_dominant[fam] = {"flops_per_launch": f_fl, "avg_launch_ms": f_ms / max(f_n, 1), "launches": f_n,
                                "achieved_tflops": (f_fl * f_n / (f_ms * 1e-3) / 1e12 if f_ms > 0 else None)}
    prof = nv.prof_collect()
    nv.prof_enable(False)
    loss = last_timed
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last = {k: float(v.item()) for k, v in loss.items()}
    if not all(math.isfinite(v) for v in last.values()):
        raise SystemExit(f"non-finite loss terms: {last}")

    def timed_option(precision, steps, what):
        """Same step with another operand precision for the Linear products; never the headline value."""
        model.set_gemm_precision(precision)
        for i in range(3):
            step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(steps):
            lb = step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb
        if world > 1:
            t = torch.tensor([tb], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tb = float(t.item())
        model.set_gemm_precision(main_precision)
        return {"metric": f"ELBO train samples/sec ({what})", "value": world * B * steps / tb, "unit": "samples/s",
                "ms_per_step": tb / steps * 1e3, "steps": steps, "last_loss": float(lb["loss"].item())}

    main_precision = model.gemm_precision
    options = {}
    if args.option_steps > 0:
        notes = {"fp32_split2": ("fp32_split2_option", "Linear products on two per-tensor-scaled fp16 pieces per value, 3 fp16 MFMAs per product"),
                 "fp32_split": ("fp32_split_option", "Linear products on three bf16 pieces per value (the fp32 values exactly), 6 bf16 MFMAs per product"),
                 "fp32": ("fp32_mfma_option", "Linear products on the fp32 MFMA instruction, v_mfma_f32_32x32x2_f32"),
                 "bf16": ("bf16_option", "bf16 GEMM operands, fp32 accumulate and fp32 everywhere else")}
        for prec, (key, what) in notes.items():
            if prec != main_precision:
                options[key] = timed_option(prec, args.option_steps, what)

    scoring = imputation = e2e = wide = None
    del pool
    model.to("cpu")
    torch.cuda.empty_cache()
    if args.scoring_profiles > 0:
        scoring = scoring_benchmark(dev, rank, world, args.scoring_profiles, args.scoring_k, args.scoring_chunk)
        torch.cuda.empty_cache()
    if world == 1 and args.imputation_rows > 0:
        imputation = imputation_benchmark(dev, args.imputation_rows, args.imputation_sweeps)
        torch.cuda.empty_cache()
    if world == 1 and args.wide_steps > 0:
        wide = wide_benchmark(dev, 4096, 16, args.wide_steps)
        nv.set_gemm_precision(main_precision)
        torch.cuda.empty_cache()
    if world == 1 and args.e2e_iterations > 0:
        e2e = e2e_benchmark(dev, args.e2e_iterations)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        gemm_ms = sum(prof[k][0] for k in ("gemm_fwd", "gemm_dgrad", "gemm_wgrad"))
        gemm_fl = sum(prof[k][1] for k in ("gemm_fwd", "gemm_dgrad", "gemm_wgrad"))
        fwd_ms, fwd_fl, _, fwd_n = prof["gemm_fwd"]
        # dominant kernel: the Linear+ReLU+column-statistics GEMM at the decoder's hidden shape (S*B x H x H)
        achieved = dom_fl * dom_n / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        peak, peak_note = _peak(main_precision)
        fwd_leg = "training_fwd_pairs" if dom_kernel == "pairs" else "training_fwd"
        o_ms, o_fl, o_n = by_kernel["plain" if dom_kernel == "pairs" else "pairs"]
        other_kernel = None
        if o_n > 0 and o_fl >= 0.999 * dom_fl:
            o_tf = o_fl * o_n / (o_ms * 1e-3) / 1e12
            other_kernel = {"kernel": (f"{KERNEL_NAME[main_precision]}<k-contig,k-contig,relu+stats,pre-split weight> (fp32 input behind a folded BatchNorm)"
                                       if dom_kernel == "pairs" else "gemm_pq256<relu+stats, pair-format input>"),
                            "avg_launch_ms": o_ms / o_n, "launches": o_n, "achieved": o_tf, "frac": o_tf / peak,
                            "traffic": recorded_traffic("training_fwd" if dom_kernel == "pairs" else "training_fwd_pairs", main_precision, S * B)[0]}
        kernels = {k: {"ms_per_step": v[0] / BREAKDOWN_STEPS, "launches_per_step": v[3] / BREAKDOWN_STEPS,
                       "tflops": (v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 and v[1] > 0 else None),
                       "gbps": (v[2] / (v[0] * 1e-3) / 1e9 if v[0] > 0 and v[2] > 0 else None)} for k, v in prof.items()}
        out = {
            "metric": "ELBO train samples/sec", "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "n_ranks_seen": n_ranks_seen,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config0.json training step, fused ELBO fwd/bwd + clip + Adam, fp32 (" + PRECISION_NOTE[main_precision] + ")",
                       "gemm_precision": main_precision, "step_replayed_from_hip_graph": graphed, "global_batch": world * B,
                       "per_gpu_batch": B, "num_mc_samples": S, "input_dim": D, "cond_dim": C, "latent_dim": 48, "hidden": 1000,
                       "vocab": 100, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": (f"gemm_pq256<relu+stats, 256 x 256 tile, pair-format input, operands by global->LDS DMA> M={S * B} N=K=1000 (decoder hidden Linear forward)"
                                                     if dom_kernel == "pairs" else
                                                     f"{KERNEL_NAME[main_precision]}<k-contig,k-contig,relu+stats{',pre-split weight' if main_precision.startswith('fp32_split') else ''}> M={S * B} N=K=1000 (decoder hidden Linear forward)"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": peak_note,
                         "fp32_mfma_peak": PEAK_F32_MFMA_TFLOPS, "vs_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                         "flops_per_launch": dom_fl, "avg_launch_ms": dom_ms / max(dom_n, 1), "launches": dom_n,
                         "measured_in": ("5 eager steps right after the timed region (the timed steps are replayed from a HIP graph)" if graphed
                                         else "the timed region"),
                         "traffic": recorded_traffic(fwd_leg, main_precision, S * B)[0],
                         "traffic_source": recorded_traffic(fwd_leg, main_precision, S * B)[1],
                         "algorithmic_bytes": 4.0 * (S * B * 1000 + 1000 * 1000 + S * B * 1000),
                         "same_shape_other_kernel": other_kernel,
                         "limiter_note": ("fp32_split2, the 128-wide kernel: with every MFMA removed it takes 0.397 ms instead of 0.408 at the full clock: "
                                          "its time is set by the global-memory path (activations re-read 8x and the weight image swept from L2, "
                                          "512-byte output row pieces), profiles/r2b_gemm_two_piece_pmc.md; the 256-wide kernel on pair-format input: matrix side "
                                          "alone 0.286 ms, memory side alone 0.226 ms, two thirds overlapped, profiles/r3_pairs_probe.md") if main_precision == "fp32_split2" else None,
                         "other_families_dominant": {k: dict(v, frac=(v["achieved_tflops"] / peak if v["achieved_tflops"] else None))
                                                     for k, v in family_dominant.items()},
                         "fwd_family_tflops": fwd_fl / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else None,
                         "all_gemm_tflops": gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None,
                         "step_model_tflops": flops_per_sample(S) * B / (ms_step * 1e-3) / 1e12,
                         "step_frac_of_fp32_mfma_peak": flops_per_sample(S) * B / (ms_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "two_piece_guard": dict(headline_guard, all_legs=guard_report()),
            "kernels": kernels,
            "kernels_note": f"per kernel family, HIP events around every launch of {BREAKDOWN_STEPS} extra steps after the timed region",
            "last_loss": last,
        }
        if scoring is not None:
            out["secondary"] = scoring
        out.update(options)
        for key, leg in (("imputation", imputation), ("wide_conditioning", wide), ("e2e", e2e)):
            if leg is not None:
                out[key] = leg
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if scoring is not None:
                scoring["cpu_baseline"] = cpu_scoring_baseline(args.scoring_k)
            if imputation is not None:
                imputation["cpu_baseline"] = cpu_imputation_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:          # a failing rank must end the job, not leave the others in a collective
        if not (isinstance(exc, SystemExit) and exc.code in (0, None)):
            import traceback
            sys.stderr.write(f"[bench rank {os.environ.get('RANK', '0')}] failed:\n{traceback.format_exc()}\n")
            sys.stderr.flush()
            os._exit(1)
        raise
