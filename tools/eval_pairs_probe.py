#!/usr/bin/env python3
"""Eval-mode unit (Linear + ReLU + folded BatchNorm) at the scoring shape, fp32 and pair-format activations on either side:
    python tools/eval_pairs_probe.py [M=819200] [N=K=1000] [reps=5]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv  # noqa: E402

P = nv.ptr


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda:0")
    nv.ensure_device(dev)
    nv.set_gemm_precision("fp32_split2")
    L = nv.lib()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 819200
    N = K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    ld = (K + 31) // 32 * 32
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.zeros(M, ld, device=dev)
    X[:, :K] = torch.randn(M, K, device=dev, generator=g).clamp_(min=0)
    W = torch.randn(N, K, device=dev, generator=g) * 0.03
    b, sc, sh = torch.randn(N, device=dev, generator=g), torch.rand(N, device=dev, generator=g) + 0.5, torch.randn(N, device=dev, generator=g) * 0.1
    xmax, wmax = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
    nv.call("btvae_absmax", P(X), X.numel(), P(xmax))
    nv.call("btvae_absmax", P(W), W.numel(), P(wmax))
    planes = torch.empty(L.btvae_weight_planes_elems(N, K), dtype=torch.int16, device=dev)
    nv.call("btvae_weight_planes_f16", P(W), K, N, K, P(planes), 1, P(wmax))
    img = torch.empty(L.btvae_weight_image_pq_bytes(N, K, 0), dtype=torch.uint8, device=dev)
    nv.call("btvae_weight_image_pq", P(W), K, N, K, 0, P(img), P(wmax))
    Xp, xs = torch.empty_like(X), torch.empty((M + 127) // 128, (K + 127) // 128, device=dev)
    nv.call("btvae_pairs_encode", P(X), ld, P(Xp), ld, P(xs), M, K)
    Y, ys = torch.empty(M, ld, device=dev), torch.empty((M + 127) // 128, (N + 127) // 128, device=dev)
    Yref = torch.empty(M, ld, device=dev)

    def f32_f32():
        L.btvae_gemm_operand_maxima(P(xmax), P(wmax))
        nv.call("btvae_linear_relu_affine_fwd_pw_ld", P(X), ld, P(W), P(planes), P(b), P(sc), P(sh), None, 0, P(Yref), ld, M, K, N)

    def f32_pairs():
        L.btvae_gemm_operand_maxima(P(xmax), P(wmax))
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(X), ld, None, P(W), P(planes), None, P(b), P(sc), P(sh), 0, P(Y), ld, P(ys), M, K, N)

    def pairs_f32():
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(Xp), ld, P(xs), None, P(img), P(wmax), P(b), P(sc), P(sh), 0, P(Y), ld, None, M, K, N)

    def pairs_pairs():
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(Xp), ld, P(xs), None, P(img), P(wmax), P(b), P(sc), P(sh), 0, P(Y), ld, P(ys), M, K, N)

    def f32_f32_res():
        L.btvae_gemm_operand_maxima(P(xmax), P(wmax))
        nv.call("btvae_linear_relu_affine_fwd_pw_ld", P(X), ld, P(W), P(planes), P(b), P(sc), P(sh), P(X), ld, P(Yref), ld, M, K, N)

    def pairs_pairs_res():
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(Xp), ld, P(xs), None, P(img), P(wmax), P(b), P(sc), P(sh), 1, P(Y), ld, P(ys), M, K, N)

    f32_f32()
    pairs_f32()
    torch.cuda.synchronize()
    print("pairs -> fp32 against fp32 -> fp32: max abs difference", (Y[:, :N] - Yref[:, :N]).abs().max().item())
    fl = 2.0 * M * N * K
    for name, fn in (("fp32 -> fp32 (shipped 128-wide kernel)", f32_f32), ("fp32 -> pairs", f32_pairs), ("pairs -> fp32", pairs_f32), ("pairs -> pairs", pairs_pairs),
                     ("fp32 -> fp32 again", f32_f32), ("pairs -> pairs again", pairs_pairs),
                     ("residual unit, fp32 -> fp32", f32_f32_res), ("residual unit, pairs -> pairs", pairs_pairs_res)):
        t = timeit(fn, reps)
        print(f"{name:42s} {t * 1e3:8.3f} ms {fl / t / 1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
