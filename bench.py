#!/usr/bin/env python3
"""Benchmark of the config0 ELBO training step on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1, one rank per GPU)

A step = one ``CVAE.train_core`` (forward, ELBO, backward, global-norm clip, Adam) on one resident batch of
synthetic daily load profiles of the config0 shape (D=48, C=4, L=48, H=1000, V=100, S=16 MC samples,
batch 4096 per GPU, fp32).  Inputs are generated on the device before the timed region.  Rank 0 prints one
JSON line; ``value`` is whole-job samples/s.  Also reported: the roofline of the dominant kernel (the fp32 MFMA
GEMM) from HIP-event timing on the launch stream inside the timed region, and a CPU baseline (the oracle
restatement timed on the host cores, rank 0, N=1 only).
"""
import argparse
import contextlib
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
# HBM-side bytes of one launch of the dominant kernel (M=65536, N=K=1000) from PMC passes kept under profiles/
DOMINANT_KERNEL_HBM_BYTES = {"fp32": 2 * 324658 * 1024 + 304764 * 1024, "fp32_split": 2 * 505912 * 1024 + 258478 * 1024}


KERNEL_NAME = {"fp32_split": "gemm_split_mfma", "fp32": "gemm_f32_mfma", "bf16": "gemm_bf16_mfma"}
PRECISION_NOTE = {
    "fp32_split": "Linear products: fp32 operands as exact 3-way bf16 splits, 6 bf16 MFMAs per product with fp32 accumulation; "
                  "error against fp64 at or below the fp32 MFMA instruction's (tools/gemm_accuracy.py)",
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


def cpu_baseline(B=1024, S=16, steps=2):
    """The CPU oracle's train step (same algorithm as the reference's CPU path) on the host cores."""
    from oracle import refimpl as R
    cores = os.cpu_count() or 1
    threads = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(threads)
    spec = R.model_spec(D, C, CONFIG0_MODEL)
    sd = R.init_state(spec, SEED)
    st = R.AdamState()
    rng = np.random.default_rng(SEED)
    g = torch.Generator().manual_seed(SEED)
    ts = []
    for i in range(steps + 1):
        x, c = synth_batch(rng, B)
        eps = torch.randn(S, B, spec.latent_dim, generator=g)
        t0 = time.perf_counter()
        R.train_step(sd, spec, st, torch.from_numpy(x), torch.from_numpy(c), eps, lr=1e-3, weight_decay=1e-5, beta=1.0, max_norm=1.0)
        ts.append(time.perf_counter() - t0)
    dt = sum(ts[1:]) / steps
    return {"value": B / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{steps} train steps of B={B}, S={S} (config0 model) after 1 warm-up step, {dt:.2f} s/step"}


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
    cores = os.cpu_count() or 1
    threads = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(threads)
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


def scoring_benchmark(dev, rank, world, n_profiles, K, chunk):
    """K-sample importance-weighted log-likelihood of the config0_gipuzkoa-shaped model (D=24, L=24, C=4, H=1000,
    V=100, eval mode): every rank scores its own ``n_profiles`` resident profiles (weak scaling), one reduce of
    (sum ll, count) at the end.  Returns whole-job profiles/s."""
    import torch.distributed as dist
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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ll, _, _ = testing_lib.iw_loglikelihood_device(model, xd, cd, K, chunk)
    tot, cnt = reduce_loglik(float(ll.double().sum().item()), n_profiles, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    flops = (5_076_000 + K * 5_152_000) * 2.0 * cnt
    return {"metric": "MC-scored profiles/sec at K=1000", "value": cnt / dt, "unit": "profiles/s", "K": K, "profiles": cnt,
            "seconds": dt, "mean_loglik": tot / cnt, "model_tflops": flops / dt / 1e12,
            "config": {"workload": "config0_gipuzkoa-shaped IW log-likelihood, eval mode", "gemm_precision": model.gemm_precision,
                       "input_dim": 24, "latent_dim": 24,
                       "hidden": 1000, "vocab": 100, "mc_chunk": chunk, "per_gpu_profiles": n_profiles}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--option-steps", "--bf16-steps", dest="option_steps", type=int, default=10,
                    help="extra train steps per alternative GEMM precision (native fp32 MFMA, bf16 operands), reported beside the headline (0 = skip)")
    ap.add_argument("--scoring-profiles", type=int, default=4096, help="per-GPU profiles of the K=1000 scoring leg (0 = skip)")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

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
    B, S = args.batch, CONFIG0_TRAIN["num_mc_samples"]
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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Inside the timed region only the dominant kernel is bracketed by HIP events (the decoder's hidden Linear
    # forward, 3 launches a step): two event records per launch cost stream time, 0.7 ms a step when every launch of
    # the step carries them.  The per-family breakdown comes from a few extra, untimed steps afterwards.
    dominant_flops = 2.0 * (S * B) * 1000 * 1000
    nv.prof_filter(["gemm_fwd"], 0.9 * dominant_flops)
    nv.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dom_ms, dom_fl, dom_n = nv.prof_dominant("gemm_fwd")
    nv.prof_collect()
    last_timed = loss
    nv.prof_filter(None, 0.0)
    nv.prof_enable(True)
    for i in range(BREAKDOWN_STEPS):
        step(i)
    torch.cuda.synchronize()
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
    native_opt = bf16_opt = None
    if args.option_steps > 0:
        native_opt = timed_option("fp32", args.option_steps, "Linear products on the fp32 MFMA instruction, v_mfma_f32_32x32x2_f32")
        bf16_opt = timed_option("bf16", args.option_steps, "bf16 GEMM operands, fp32 accumulate and fp32 everywhere else")

    scoring = None
    if args.scoring_profiles > 0:
        del pool
        model.to("cpu")
        torch.cuda.empty_cache()
        scoring = scoring_benchmark(dev, rank, world, args.scoring_profiles, args.scoring_k, args.scoring_chunk)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        gemm_ms = sum(prof[k][0] for k in ("gemm_fwd", "gemm_dgrad", "gemm_wgrad"))
        gemm_fl = sum(prof[k][1] for k in ("gemm_fwd", "gemm_dgrad", "gemm_wgrad"))
        fwd_ms, fwd_fl, _, fwd_n = prof["gemm_fwd"]
        # dominant kernel: the Linear+ReLU+column-statistics GEMM at the decoder's hidden shape (S*B x H x H)
        achieved = dom_fl * dom_n / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        if main_precision == "fp32_split":
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_TERMS
            peak_note = "fp32-equivalent FLOP/s: bf16 dense peak 2500 / 6 bf16 MFMAs per fp32 product"
        else:
            peak = PEAK_F32_MFMA_TFLOPS if main_precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
            peak_note = "dense matrix peak of the operand type"
        kernels = {k: {"ms_per_step": v[0] / BREAKDOWN_STEPS, "launches_per_step": v[3] / BREAKDOWN_STEPS,
                       "tflops": (v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 and v[1] > 0 else None),
                       "gbps": (v[2] / (v[0] * 1e-3) / 1e9 if v[0] > 0 and v[2] > 0 else None)} for k, v in prof.items()}
        out = {
            "metric": "ELBO train samples/sec", "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config0.json training step, fused ELBO fwd/bwd + clip + Adam, fp32 (" + PRECISION_NOTE[main_precision] + ")",
                       "gemm_precision": main_precision, "global_batch": world * B,
                       "per_gpu_batch": B, "num_mc_samples": S, "input_dim": D, "cond_dim": C, "latent_dim": 48, "hidden": 1000,
                       "vocab": 100, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": f"{KERNEL_NAME[main_precision]}<k-contig,k-contig,relu+stats{',pre-split weight' if main_precision == 'fp32_split' else ''}> M={S * B} N=K=1000 (decoder hidden Linear forward)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": peak_note,
                         "fp32_mfma_peak": PEAK_F32_MFMA_TFLOPS, "vs_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                         "flops_per_launch": dom_fl, "avg_launch_ms": dom_ms / max(dom_n, 1), "launches": dom_n,
                         "traffic": DOMINANT_KERNEL_HBM_BYTES.get(main_precision) if (S * B == 65536) else None,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE of this kernel at this shape, separate passes: profiles/r1_gemm_fwd_pmc.md (not collected live)",
                         "algorithmic_bytes": 4.0 * (S * B * 1000 + 1000 * 1000 + S * B * 1000),
                         "fwd_family_tflops": fwd_fl / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else None,
                         "all_gemm_tflops": gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None,
                         "step_model_tflops": flops_per_sample(S) * B / (ms_step * 1e-3) / 1e12,
                         "step_frac_of_fp32_mfma_peak": flops_per_sample(S) * B / (ms_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "kernels": kernels,
            "kernels_note": f"per kernel family, HIP events around every launch of {BREAKDOWN_STEPS} extra steps after the timed region",
            "last_loss": last,
        }
        if scoring is not None:
            out["secondary"] = scoring
        if native_opt is not None:
            out["fp32_mfma_option"] = native_opt
        if bf16_opt is not None:
            out["bf16_option"] = bf16_opt
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if scoring is not None:
                scoring["cpu_baseline"] = cpu_scoring_baseline(args.scoring_k)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
