#!/usr/bin/env python3
"""Error of a Linear product against an fp64 product of the same fp32 operands, per GEMM precision mode: rms and max
error relative to the rms magnitude of the exact result.  The product is the dX one (C[M,Kout] = A[M,R] W[R,Kout], plain
epilogue: every mode has it), at reduction lengths R = 52 and 1000, M = 4096, Kout = 1000, for Gaussian operands,
heavy-tailed ones (cubed Gaussians) and Gaussian ones with a few elements 2^12 and 2^20 above the rest (the case the
per-tensor scale of the two-piece fp16 mode is weakest at)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv  # noqa: E402

P = nv.ptr


def main():
    dev = torch.device("cuda:0")
    nv.ensure_device(dev)
    M, KO = 4096, 1000
    g = torch.Generator().manual_seed(0)
    for R in (52, 1000):
        A0, W0 = torch.randn(M, R, generator=g), torch.randn(R, KO, generator=g) * 0.03
        out12, out20 = A0.clone(), A0.clone()
        out12[::97, ::13] *= 2.0 ** 12
        out20[5, 7] = 2.0 ** 20
        cases = (("gaussian", A0, W0), ("heavy-tailed", A0 ** 3, torch.randn(R, KO, generator=g) ** 3 * 0.03),
                 ("outliers 2^12", out12, W0), ("one 2^20", out20, W0))
        for label, A, W in cases:
            ref = A.double() @ W.double()
            # rows without an outlier are where a per-tensor scale could hurt: report them apart
            plain = A.abs().max(1).values < 100
            scale_all = ref.pow(2).mean().sqrt().item()
            scale_plain = ref[plain].pow(2).mean().sqrt().item()
            Ad, Wd = A.to(dev), W.to(dev)
            for mode in ("fp32", "fp32_split", "fp32_split2", "bf16"):
                nv.set_gemm_precision(mode)
                C = torch.empty(M, KO, device=dev)
                if mode == "fp32_split2":
                    ma, mw = torch.zeros(2, device=dev), torch.zeros(2, device=dev)
                    nv.call("btvae_absmax", P(Ad), Ad.numel(), P(ma))
                    nv.call("btvae_absmax", P(Wd), Wd.numel(), P(mw))
                    nv.lib().btvae_gemm_operand_maxima(P(ma), P(mw))
                nv.call("btvae_linear_dgrad", P(Ad), P(Wd), None, P(C), KO, M, R, KO)
                torch.cuda.synchronize()
                err = C.cpu().double() - ref
                line = (f"R={R:5d} {label:14s} {mode:12s} rms err / rms = {err.pow(2).mean().sqrt().item() / scale_all:.3e}   "
                        f"max err / rms = {err.abs().max().item() / scale_all:.3e}")
                if not bool(plain.all()):
                    e = err[plain]
                    line += f"   rows without outliers: rms {e.pow(2).mean().sqrt().item() / scale_plain:.3e}  max {e.abs().max().item() / scale_plain:.3e}"
                print(line)
            nv.set_gemm_precision("fp32")


if __name__ == "__main__":
    main()
