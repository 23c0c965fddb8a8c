#!/usr/bin/env python3
"""Error of the Linear-forward product against an fp64 product of the same fp32 operands, per GEMM precision mode:
rms and max error relative to the rms magnitude of the exact result (M=4096, N=K=1000, N(0,1) activations,
N(0, 0.03) weights; also heavy-tailed operands)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv  # noqa: E402

P = nv.ptr


def main():
    dev = torch.device("cuda:0")
    nv.ensure_device(dev)
    M, N, K = 4096, 1000, 1000
    g = torch.Generator().manual_seed(0)
    for label, X, W in (("gaussian", torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.03),
                        ("heavy-tailed", torch.randn(M, K, generator=g) ** 3, torch.randn(N, K, generator=g) ** 3 * 0.03)):
        b = torch.zeros(N)
        ref = X.double() @ W.double().t()
        scale = ref.pow(2).mean().sqrt().item()
        Xd, Wd, bd = X.to(dev), W.to(dev), b.to(dev) - 1e30      # bias -1e30 would clip; use the head epilogue instead
        for mode in ("fp32", "fp32_split", "bf16"):
            nv.set_gemm_precision(mode)
            Y = torch.empty(M, N, device=dev)
            nv.call("btvae_linear_head_fwd", P(Xd), K, P(Wd), P(b.to(dev)), 0, -1e30, 1e30, P(Y), None, None, M, K, N)
            torch.cuda.synchronize()
            err = (Y.cpu().double() - ref)
            print(f"{label:13s} {mode:10s} rms err / rms = {err.pow(2).mean().sqrt().item() / scale:.3e}   max err / rms = {err.abs().max().item() / scale:.3e}")
        nv.set_gemm_precision("fp32")


if __name__ == "__main__":
    main()
