#!/usr/bin/env python3
"""bf16-stored activations: the 256 x 256 / 128 x 256 DMA-fed forward product (bf16 in, bf16 out) against fp64 on the
bf16-rounded operands, and its time next to the shipped bf16-operand kernel on fp32 tensors.  Needs the experiments build
(see tools/pairs_probe.py)."""
import os
import sys
import time
from ctypes import c_int as I, c_longlong, c_void_p as VP

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv  # noqa: E402

P = nv.ptr
L = nv.lib()
L.btvae_weight_image_bf16.argtypes, L.btvae_weight_image_bf16.restype = [VP, I, I, I, I, VP, VP], I
L.btvae_weight_image_bf16_bytes.argtypes, L.btvae_weight_image_bf16_bytes.restype = [I, I, I], c_longlong
L.btvae_linear_relu_stats_fwd_bh256.argtypes, L.btvae_linear_relu_stats_fwd_bh256.restype = [VP, I, VP, VP, VP, I, VP, I, I, I, I, VP], I


def timeit(fn, reps):
    for _ in range(5):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    N = K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ld = (K + 31) // 32 * 32
    g = torch.Generator(device=dev).manual_seed(0)
    for data in ("relu", "zero"):
        X = torch.zeros(M, ld, device=dev)
        if data != "zero":
            X[:, :K] = torch.randn(M, K, device=dev, generator=g).clamp_(min=0)
        W = torch.randn(N, K, device=dev, generator=g) * (0.0 if data == "zero" else 0.03)
        b = torch.randn(N, device=dev, generator=g)
        Xb = X.to(torch.bfloat16)
        Yb = torch.full((M, ld), float("nan"), device=dev, dtype=torch.bfloat16)
        Y = torch.empty(M, ld, device=dev)
        part, part2 = torch.empty((M + 127) // 128, 2, N, device=dev), torch.zeros((M + 127) // 128, 2, N, device=dev)
        img = torch.empty(L.btvae_weight_image_bf16_bytes(N, K, 0), dtype=torch.uint8, device=dev)
        nv.call("btvae_weight_image_bf16", P(W), K, N, K, 0, P(img))

        def base():
            L.btvae_next_row_pitches(ld, 0)
            nv.call("btvae_linear_relu_stats_fwd", P(X), ld, P(W), P(b), P(Y), P(part), M, K, N)

        def bh(groups):
            def run():
                nv.call("btvae_linear_relu_stats_fwd_bh256", P(Xb), ld, P(img), P(b), P(Yb), ld, P(part2), M, K, N, groups)
            return run

        nv.set_gemm_precision("bf16")
        base()
        rows = slice(0, min(M, 2048))
        ref = torch.relu(Xb[rows, :K].double() @ W.to(torch.bfloat16).double().t() + b.double())
        for groups in (2, 1):
            Yb.fill_(float("nan")); part2.zero_()
            bh(groups)()
            torch.cuda.synchronize()
            err = (Yb[rows, :N].double() - ref).abs().max().item()
            pad_ok = bool((Yb[:, N:] == 0).all())
            print(f"[{data}] groups {groups}: max abs err vs fp64 of the bf16 operands {err:.3e} (bf16 output rounding ~{ref.abs().max().item() * 2 ** -9:.2e}); "
                  f"pad columns zero: {pad_ok}; stats vs shipped kernel max diff {(part2 - part).abs().max().item():.3e}")
        for _ in range(10):
            base()
        fl = 2.0 * M * N * K
        t = timeit(base, reps)
        print(f"[{data}] shipped bf16-operand kernel, fp32 tensors: {t*1e3:.3f} ms {fl/t/1e12:.1f} TFLOP/s")
        for groups in (2, 1):
            t = timeit(bh(groups), reps)
            print(f"[{data}] bf16 in / bf16 out, DMA, {128 * groups} x 256 tile: {t*1e3:.3f} ms {fl/t/1e12:.1f} TFLOP/s")
        nv.set_gemm_precision("fp32")


if __name__ == "__main__":
    main()
