#!/usr/bin/env python3
"""Micro-benchmark of the GEMM family at the config0 decoder shapes (M=65536, N=K=1000): prints TFLOP/s per
layout.  Used on its own and under ``rocprofv3 --pmc`` to read MFMA-busy / wait counters."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv  # noqa: E402

P = nv.ptr


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    nv.ensure_device(dev)
    if os.environ.get("BTVAE_BENCH_BF16"):
        nv.set_gemm_precision("bf16")
    if os.environ.get("BTVAE_BENCH_PREC"):
        nv.set_gemm_precision(os.environ["BTVAE_BENCH_PREC"])
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    N = K = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    if os.environ.get("BTVAE_BENCH_K"):             # reduction length on its own: forward products only (the other entries
        K = int(os.environ["BTVAE_BENCH_K"])        # of `ops` assume N == K in how their buffers are sized and are not run)
        if which not in ("fwd", "fwd_pw"):
            raise SystemExit("BTVAE_BENCH_K applies to fwd / fwd_pw only")
    g = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.03
    b = torch.randn(N, device=dev, generator=g)
    if os.environ.get("BTVAE_BENCH_ZERO"):          # all-zero operands: the rate the instruction stream allows when no datapath toggles
        X.zero_(); W.zero_()
    ldx = int(os.environ.get("BTVAE_BENCH_LDX", K))        # row pitch of the forward products' activation operand
    Xf = X                                                 # the same values in a buffer of that pitch (dense [M, ldx])
    if ldx != K:
        Xf = torch.zeros(M, ldx, device=dev)
        Xf[:, :K] = X
    Y = torch.empty(M, N, device=dev)
    part = torch.empty((M + 127) // 128, 2, max(N, K), device=dev)
    dX = torch.empty(M, K, device=dev)
    dW = torch.empty(N, K, device=dev)
    ws = torch.empty(32 * 1024 * 1024, device=dev)
    two = nv.get_gemm_precision() == "fp32_split2"
    mx = {}

    def amax(t):
        if id(t) not in mx:
            mx[id(t)] = torch.zeros(2, device=dev)
            nv.call("btvae_absmax", P(t), t.numel(), P(mx[id(t)]))
        return mx[id(t)]

    def make_planes(Wm, n, k):
        pl = torch.empty(nv.lib().btvae_weight_planes_elems(n, k), dtype=torch.int16, device=dev)
        if two:
            nv.call("btvae_weight_planes_f16", P(Wm), k, n, k, P(pl), 3, P(amax(Wm)))
        else:
            nv.call("btvae_weight_planes", P(Wm), k, n, k, P(pl), 3)
        return pl

    def mxd(fn, a, b):
        """``fn`` preceded, in the two-piece mode, by the maxima of its A and B operand tensors."""
        if not two:
            return fn
        ma, mb = amax(a), amax(b)

        def run():
            nv.lib().btvae_gemm_operand_maxima(P(ma), P(mb))
            fn()
        return run

    planes = make_planes(W, N, K)
    # head-output shapes (reduction over V=100 outputs): epilogue- and HBM-heavy dX product fused with the producing
    # unit's BatchNorm backward, and the head forward
    V = 100
    Wh = torch.randn(V, K, device=dev, generator=g) * 0.03
    dV = torch.randn(M, V, device=dev, generator=g)
    hp = make_planes(Wh, V, K)
    mean, inv, gam, sums = (torch.rand(K, device=dev, generator=g) + 0.5 for _ in range(4))
    sums2 = torch.randn(2, K, device=dev, generator=g)
    colpart = torch.empty((M + 127) // 128, K, device=dev)
    hv, hd, hs = (torch.empty(M, V, device=dev) for _ in range(3))
    ops = {
        "headout_bnbwd": lambda: nv.call("btvae_linear_dgrad_bnbwd_pw", P(dV), P(Wh), P(hp), None, P(X), P(mean), P(inv), P(gam), P(sums2), M, P(dX), P(colpart), M, V, K),
        "headout_plain": lambda: nv.call("btvae_linear_dgrad_pw", P(dV), P(Wh), P(hp), None, P(dX), K, M, V, K),
        "headout_gstats": lambda: nv.call("btvae_linear_dgrad_gstats_pw", P(dV), P(Wh), P(hp), None, P(X), P(mean), P(inv), P(dX), P(part), M, V, K),
        "gstats_big": lambda: nv.call("btvae_linear_dgrad_gstats_pw", P(Y), P(W), P(planes), P(X), P(X), P(mean), P(inv), P(dX), P(part), M, N, K),
        "bnbwd_big": lambda: nv.call("btvae_linear_dgrad_bnbwd_pw", P(Y), P(W), P(planes), P(X), P(X), P(mean), P(inv), P(gam), P(sums2), M, P(dX), P(colpart), M, N, K),
        "head_fwd": lambda: nv.call("btvae_linear_head_fwd", P(X), K, P(Wh), P(b), 2, 1e-4, 0.62, P(hv), P(hd), P(hs), M, K, V),
        "fwd_pw": lambda: nv.call("btvae_linear_relu_stats_fwd_pw", P(Xf), ldx, P(W), P(planes), P(b), P(Y), P(part), M, K, N),
        "dgrad_pw": lambda: nv.call("btvae_linear_dgrad_pw", P(Y), P(W), P(planes), None, P(dX), K, M, N, K),
        "fwd": lambda: nv.call("btvae_linear_relu_stats_fwd", P(X), K, P(W), P(b), P(Y), P(part), M, K, N),
        "dgrad": lambda: nv.call("btvae_linear_dgrad", P(Y), P(W), None, P(dX), K, M, N, K),
        "wgrad": lambda: nv.call("btvae_linear_wgrad", P(Y), P(X), K, P(dW), P(ws), ws.numel(), M, N, K, 0),
    }
    Y.normal_(generator=g)         # the dX entries read Y as their A operand before any forward entry has written it
    a_of = {"headout_bnbwd": (dV, Wh), "headout_plain": (dV, Wh), "headout_gstats": (dV, Wh), "gstats_big": (Y, W), "bnbwd_big": (Y, W),
            "fwd_pw": (Xf, W), "dgrad_pw": (Y, W), "fwd": (X, W), "dgrad": (Y, W), "wgrad": (Y, X)}
    if N != K:
        ops = {k: v for k, v in ops.items() if k in ("fwd", "fwd_pw")}
    if which != "all":             # run (and warm up with) the selected entry only: a forward entry would rewrite Y, whose
        ops = {k: v for k, v in ops.items() if k == which}      # maximum the dX entries were given
    elif two:
        ops = {k: v for k, v in ops.items() if not k.startswith("fwd") and k != "head_fwd"}
    ops = {k: (mxd(v, *a_of[k]) if k in a_of else v) for k, v in ops.items()}
    for _ in range(15):          # let the clock settle before anything is timed
        for fn in ops.values():
            fn()
    torch.cuda.synchronize()
    for name, fn in ops.items():
        if which not in ("all", name):
            continue
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
