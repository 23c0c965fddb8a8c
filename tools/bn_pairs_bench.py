"""BatchNorm-apply pass at the training shape (M = 65,536, N = 1000, pitch 1024): the streaming fp32 pass against the
block-wise pass that writes the pair format, with and without a residual add-in (HIP events over 50 launches each).

    python tools/bn_pairs_bench.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv

dev = torch.device("cuda:0")
nv.ensure_device(dev)
P = nv.ptr
M, N, ld = 65536, 1000, 1024
Y = torch.randn(M, ld, device=dev).relu_()
R = torch.randn(M, ld, device=dev)
Y[:, N:] = 0; R[:, N:] = 0
mean, invstd, gamma, beta = (torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5, torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1)
out = torch.empty(M, ld, device=dev)
sc = torch.empty(M // 128, 8, device=dev)
Rp, rsc = torch.empty(M, ld, device=dev), torch.empty(M // 128, 8, device=dev)
nv.call("btvae_pairs_encode", P(R), ld, P(Rp), ld, P(rsc), M, N)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def stream(res):
    nv.lib().btvae_next_row_pitches(ld, 0)
    nv.call("btvae_bn_apply_fwd", P(Y), P(mean), P(invstd), P(gamma), P(beta), P(R) if res else None, None, 1.0, P(out), M, N)


def pairs(res):
    r, r_sc = (None, None) if res == 0 else ((P(R), None) if res == 1 else (P(Rp), P(rsc)))
    nv.call("btvae_bn_apply_fwd_pairs", P(Y), ld, P(mean), P(invstd), P(gamma), P(beta), r, ld if res else 0, r_sc, P(out), ld, P(sc), M, N)


for name, fn, mats in (("streaming, no add-in", lambda: stream(False), 2), ("streaming, fp32 add-in", lambda: stream(True), 3),
                       ("pairs out, no add-in", lambda: pairs(0), 2), ("pairs out, fp32 add-in", lambda: pairs(1), 3),
                       ("pairs out, pairs add-in", lambda: pairs(2), 3)):
    ms = timed(fn)
    print(f"{name:26s} {ms:.4f} ms  {mats * M * N * 4 / ms / 1e9:.2f} TB/s")
