#!/usr/bin/env python3
"""Pair-format activations (csrc/pairs.h): round trip, forward products against fp64, and their time next to the
fp32-operand two-piece product at the config0 decoder shape.  Needs the experiments build of the library:

    make -C behavio_temporal_vae_amd/csrc experiments [EXP_DEFS=-DBTVAE_PQ_ABLATE]
    BTVAE_HIP_LIB=$PWD/tools/libbtvae_exp.so python tools/pairs_probe.py [M] [N=K] [reps]

With BTVAE_PQ_ABLATE built in, the 256 x 256 kernel is also timed without its loads / matrix instructions / epilogue."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from behavio_temporal_vae_amd import _native as nv  # noqa: E402

P = nv.ptr
from ctypes import c_int as I, c_longlong, c_void_p as VP  # noqa: E402

for _name, _args in {"btvae_pairs_encode": [VP, I, VP, I, VP, I, I, VP], "btvae_pairs_decode": [VP, I, VP, VP, I, I, I, VP],
                     "btvae_linear_relu_stats_fwd_pq": [VP, I, VP, VP, VP, VP, VP, VP, I, I, I, I, VP],
                     "btvae_linear_relu_stats_fwd_pq256": [VP, I, VP, VP, VP, VP, VP, VP, I, I, I, I, VP],
                     "btvae_weight_image_pq": [VP, I, I, I, I, VP, VP, VP]}.items():
    _fn = getattr(nv.lib(), _name)
    _fn.argtypes, _fn.restype = _args, I
nv.lib().btvae_weight_image_pq_bytes.argtypes, nv.lib().btvae_weight_image_pq_bytes.restype = [I, I, I], c_longlong


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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    N = K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ld = (K + 31) // 32 * 32
    g = torch.Generator(device=dev).manual_seed(0)
    for data in ("relu", "randn", "zero"):
        X = torch.zeros(M, ld, device=dev)
        if data != "zero":
            X[:, :K] = torch.randn(M, K, device=dev, generator=g)
        if data == "relu":
            X.clamp_(min=0)
        W = torch.randn(N, K, device=dev, generator=g) * 0.03
        if data == "zero":
            W.zero_()
        b = torch.randn(N, device=dev, generator=g)
        Y = torch.empty(M, ld, device=dev)
        Y2 = torch.empty(M, ld, device=dev)
        part = torch.empty((M + 127) // 128, 2, N, device=dev)
        Xp = torch.empty_like(X)
        sc = torch.empty((M + 127) // 128, (K + 127) // 128, device=dev)
        nv.call("btvae_pairs_encode", P(X), ld, P(Xp), ld, P(sc), M, K)
        Xd = torch.zeros_like(X)
        nv.call("btvae_pairs_decode", P(Xp), ld, P(sc), P(Xd), ld, M, K)
        err = (Xd[:, :K] - X[:, :K]).abs().max().item()
        print(f"[{data}] round trip max abs err {err:.3e} (max |x| {X.abs().max().item():.3f})")
        wmax = torch.zeros(2, device=dev)
        xmax = torch.zeros(2, device=dev)
        nv.call("btvae_absmax", P(W), W.numel(), P(wmax))
        nv.call("btvae_absmax", P(X), X.numel(), P(xmax))
        pl = torch.empty(nv.lib().btvae_weight_planes_elems(N, K), dtype=torch.int16, device=dev)
        nv.call("btvae_weight_planes_f16", P(W), K, N, K, P(pl), 3, P(wmax))

        def base():
            nv.lib().btvae_gemm_operand_maxima(P(xmax), P(wmax))
            nv.lib().btvae_next_row_pitches(ld, 0)
            nv.call("btvae_linear_relu_stats_fwd_pw", P(X), ld, P(W), P(pl), P(b), P(Y), P(part), M, K, N)

        def pq(variant):
            def run():
                nv.lib().btvae_next_row_pitches(ld, 0)
                nv.call("btvae_linear_relu_stats_fwd_pq", P(Xp), ld, P(sc), P(pl), P(wmax), P(b), P(Y2), P(part), M, K, N, variant)
            return run

        img = torch.empty(nv.lib().btvae_weight_image_pq_bytes(N, K, 0), dtype=torch.uint8, device=dev)
        nv.call("btvae_weight_image_pq", P(W), K, N, K, 0, P(img), P(wmax))

        def pq256(groups=2):
            nv.lib().btvae_next_row_pitches(ld, 0)
            nv.call("btvae_linear_relu_stats_fwd_pq256", P(Xp), ld, P(sc), P(img), P(wmax), P(b), P(Y2), P(part2), M, K, N, groups)

        def pq256_1():
            pq256(1)

        part2 = torch.zeros_like(part)
        base()
        Y2.zero_()
        pq256()
        torch.cuda.synchronize()
        rows = slice(0, min(M, 2048))
        ref = torch.relu(X[rows, :K].double() @ W.double().t() + b.double())
        print(f"[{data}] pq256: max abs err vs fp64 {(Y2[rows, :N].double() - ref).abs().max().item():.3e}; vs base max diff "
              f"{(Y2[:, :N] - Y[:, :N]).abs().max().item():.3e}; stats max diff {(part2 - part).abs().max().item():.3e}")
        Y2.zero_(); part2.zero_()
        pq256(1)
        torch.cuda.synchronize()
        print(f"[{data}] pq256 one group: max abs err vs fp64 {(Y2[rows, :N].double() - ref).abs().max().item():.3e}; vs base max diff "
              f"{(Y2[:, :N] - Y[:, :N]).abs().max().item():.3e}; stats max diff {(part2 - part).abs().max().item():.3e}")
        for variant in (0, 1):
            Y2.zero_()
            pq(variant)()
            torch.cuda.synchronize()
            rows = slice(0, min(M, 2048))
            ref = torch.relu(X[rows, :K].double() @ W.double().t() + b.double())
            e_pq = (Y2[rows, :N].double() - ref).abs().max().item()
            e_b = (Y[rows, :N].double() - ref).abs().max().item()
            print(f"[{data}] variant {variant}: max abs err vs fp64: pairs {e_pq:.3e}, fp32-operand two-piece {e_b:.3e}; "
                  f"pairs vs base max diff {(Y2[:, :N] - Y[:, :N]).abs().max().item():.3e}")
        for _ in range(10):
            base()
        tb = timeit(base, reps)
        fl = 2.0 * M * N * K
        print(f"[{data}] fwd_pw       : {tb*1e3:.3f} ms {fl/tb/1e12:.1f} TFLOP/s")
        for variant in (0, 1):
            t = timeit(pq(variant), reps)
            print(f"[{data}] fwd_pq var {variant} : {t*1e3:.3f} ms {fl/t/1e12:.1f} TFLOP/s")
        t = timeit(pq256, reps)
        print(f"[{data}] fwd_pq256    : {t*1e3:.3f} ms {fl/t/1e12:.1f} TFLOP/s")
        t = timeit(pq256_1, reps)
        print(f"[{data}] fwd_pq256 1grp: {t*1e3:.3f} ms {fl/t/1e12:.1f} TFLOP/s")
        if os.environ.get("BTVAE_PQ_ABLATE_RUN"):
            for ab in (1, 2, 4, 5, 6, 3, 7):
                os.environ["BTVAE_PQ_ABLATE"] = str(ab)
                t = timeit(pq256, reps)
                t1 = timeit(pq256_1, reps)
                print(f"[{data}] fwd_pq256 ablate {ab} (1 no loads, 2 no mfma, 4 no epilogue): {t*1e3:.3f} ms; one group {t1*1e3:.3f} ms")
            os.environ["BTVAE_PQ_ABLATE"] = "0"


if __name__ == "__main__":
    main()
