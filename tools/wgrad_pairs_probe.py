#!/usr/bin/env python3
"""Weight gradient dW = dY^T X with X stored in the pair format (csrc/pairs.h) against the shipped two-piece product on an
fp32 X: agreement, error against fp64, time.

    python tools/wgrad_pairs_probe.py [M=65536] [N=K=1000] [reps=20]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv  # noqa: E402

P = nv.ptr


def timeit(fn, reps):
    for _ in range(3):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    N = K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ld = (K + 31) // 32 * 32
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.zeros(M, ld, device=dev)
    X[:, :K] = torch.randn(M, K, device=dev, generator=g).clamp_(min=0) * torch.logspace(-1, 1, K, device=dev)
    dY = torch.zeros(M, ld, device=dev)
    dY[:, :N] = torch.randn(M, N, device=dev, generator=g) * 1e-3
    xmax, ymax = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
    nv.call("btvae_absmax", P(X), X.numel(), P(xmax))
    nv.call("btvae_absmax", P(dY), dY.numel(), P(ymax))
    Xp, xs = torch.empty_like(X), torch.empty((M + 127) // 128, (K + 127) // 128, device=dev)
    nv.call("btvae_pairs_encode", P(X), ld, P(Xp), ld, P(xs), M, K)
    ws = torch.empty(64 * 1024 * 1024, device=dev)
    dW0, dW1 = torch.empty(N, K, device=dev), torch.empty(N, K, device=dev)

    def shipped():
        L.btvae_gemm_operand_maxima(P(ymax), P(xmax))
        L.btvae_next_row_pitches(ld, 0)
        nv.call("btvae_linear_wgrad", P(dY), P(X), ld, P(dW0), P(ws), ws.numel(), M, N, K, 0)

    def pairs():
        L.btvae_gemm_operand_maxima(P(ymax), None)
        L.btvae_next_row_pitches(ld, 0)
        nv.call("btvae_linear_wgrad_pairs", P(dY), P(Xp), ld, P(xs), P(dW1), P(ws), ws.numel(), M, N, K, 0)

    shipped()
    pairs()
    torch.cuda.synchronize()
    rows = slice(0, min(N, 64))
    ref = dY[:, rows].double().t() @ X[:, :K].double()
    scale = float(ref.abs().max())
    print(f"max |pairs - shipped| / max|dW| = {float((dW1 - dW0).abs().max()) / scale:.3e}; against fp64 (first {rows.stop} rows): "
          f"shipped {float((dW0[rows].double() - ref).abs().max()) / scale:.3e}, pairs {float((dW1[rows].double() - ref).abs().max()) / scale:.3e}")
    fl = 2.0 * M * N * K
    for name, fn in (("fp32 X (shipped)", shipped), ("pair-format X", pairs), ("fp32 X again", shipped), ("pair-format X again", pairs)):
        t = timeit(fn, reps)
        print(f"{name:24s} {t * 1e3:8.3f} ms {fl / t / 1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
