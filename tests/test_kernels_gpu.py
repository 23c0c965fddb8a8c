"""GPU: every C-ABI kernel entry point against an fp64 torch/CPU restatement of the same op.
Tolerances are fp32 round-off for the reduction length involved (stated per test)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from behavio_temporal_vae_amd import _native as nv  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    d = torch.device("cuda:0")
    nv.ensure_device(d)
    return d


@pytest.fixture(params=["fp32", "fp32_split"])
def gemm_mode(request):
    """Both fp32-accurate precisions of the Linear products: the fp32 MFMA instruction and the exact three-way bf16
    split on the bf16 matrix pipe.  Same tolerances for both."""
    nv.set_gemm_precision(request.param)
    yield request.param
    nv.set_gemm_precision("fp32")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def close(got, ref, rtol, atol=0.0, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    err = (got - ref).abs()
    lim = atol + rtol * ref.abs()
    bad = err > lim
    assert not bad.any(), f"{what}: max err {err.max().item():.3e}, worst ratio {(err / (lim + 1e-300)).max().item():.2f}"


P = nv.ptr


@pytest.mark.parametrize("M,K,N", [(300, 52, 1000), (256, 1000, 1000), (129, 40, 72), (64, 7, 5)])
def test_linear_relu_stats(dev, gemm_mode, M, K, N):
    X, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    Xd, Wd, bd = (t.float().to(dev) for t in (X, W, b))
    Y = torch.empty(M, N, device=dev)
    tiles = (M + 127) // 128
    part = torch.empty(tiles, 2, N, device=dev)
    nv.call("btvae_linear_relu_stats_fwd", P(Xd), K, P(Wd), P(bd), P(Y), P(part), M, K, N)
    ref = torch.relu(Xd.cpu().double() @ Wd.cpu().double().t() + bd.cpu().double())
    close(Y, ref, 2e-6, 2e-6 * math.sqrt(K), "y")
    mean, invstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    nv.call("btvae_bn_stats_finalize", P(part), tiles, M, N, P(mean), P(invstd), P(rm), P(rv), 0.1, 1e-5)
    yr = Y.cpu().double()
    close(mean, yr.mean(0), 1e-5, 1e-6, "mean")
    close(invstd, 1 / torch.sqrt(yr.var(0, unbiased=False) + 1e-5), 2e-5, 0, "invstd")
    close(rm, 0.1 * yr.mean(0), 1e-5, 1e-6, "running_mean")
    close(rv, 0.9 + 0.1 * yr.var(0, unbiased=True), 2e-5, 0, "running_var")


def test_row_pitches_match_dense(dev, gemm_mode):
    """``btvae_next_row_pitches``: every entry point that honours it gives, on activation / gradient matrices kept in
    buffers with padded rows, the bits it gives on dense ones; the padding of the outputs is not written; the request
    is consumed by one launch; a pitch below the width is refused."""
    M, N, K, LN, LK = 300, 200, 136, 224, 160
    L = nv.lib()
    f = lambda *shape, seed: rnd(*shape, seed=seed).float().to(dev)
    X, W, b = f(M, K, seed=1), f(N, K, seed=2) * 0.3, f(N, seed=3)
    dY, Rn, Rk = f(M, N, seed=4), f(M, N, seed=5), f(M, K, seed=6)
    Yn, Yk = torch.relu(f(M, N, seed=7)), torch.relu(f(M, K, seed=8))
    mean_n, inv_n, gam_n, bet_n = f(N, seed=9), f(N, seed=10).abs() + 0.5, f(N, seed=11), f(N, seed=12)
    mean_k, inv_k, gam_k = f(K, seed=13), f(K, seed=14).abs() + 0.5, f(K, seed=15)
    sums_n, sums_k = f(2, N, seed=16), f(2, K, seed=17)
    keep = (torch.rand(M, N, generator=torch.Generator().manual_seed(18)) > 0.3).to(torch.uint8).to(dev)
    sc_k, sh_k, dbc = f(K, seed=19), f(K, seed=20), f(N, seed=21)
    tiles, chunks = (M + 127) // 128, L.btvae_strip_chunks(M)
    ws = torch.empty(4 * 1024 * 1024, device=dev)

    def run(pitched):
        ln, lk = (LN, LK) if pitched else (N, K)

        alive = []                         # a temporary handed to P() would be freed, and its block reused, before the launch

        def buf(t, ld):                    # `t` in a buffer with rows `ld` apart, padding set to a sentinel
            o = torch.full((t.shape[0], ld), 7.5, device=dev)
            o[:, :t.shape[1]] = t
            alive.append(o)
            return o

        def out(cols, ld):
            return torch.full((M, ld), -3.25, device=dev)

        def pitches(n=True, k=False):
            if pitched:
                L.btvae_next_row_pitches(ln if n else 0, lk if k else 0)

        res = {}
        Y, part = out(N, ln), torch.zeros(tiles, 2, N, device=dev)
        pitches(); nv.call("btvae_linear_relu_stats_fwd_pw", P(X), K, P(W), None, P(b), P(Y), P(part), M, K, N)
        res["relu_stats"] = (Y, N); res["relu_stats.part"] = (part, None)
        O = out(N, ln)
        pitches(); nv.call("btvae_bn_apply_fwd", P(buf(Yn, ln)), P(mean_n), P(inv_n), P(gam_n), P(bet_n), P(buf(Rn, ln)), P(keep), 1.25, P(O), M, N)
        res["bn_apply"] = (O, N)
        D = out(K, lk)
        pitches(); nv.call("btvae_linear_dgrad_pw", P(buf(dY, ln)), P(W), None, P(buf(Rk, lk)), P(D), lk, M, N, K)
        res["dgrad"] = (D, K)
        D2, colpart = out(K, lk), torch.zeros(tiles, K, device=dev)
        pitches(True, True)
        nv.call("btvae_linear_dgrad_bnbwd_pw", P(buf(dY, ln)), P(W), None, P(buf(Rk, lk)), P(buf(Yk, lk)), P(mean_k), P(inv_k), P(gam_k), P(sums_k), M,
                P(D2), P(colpart), M, N, K)
        res["bnbwd"] = (D2, K); res["bnbwd.colpart"] = (colpart, None)
        D3, gpart = out(K, lk), torch.zeros(tiles, 2, K, device=dev)
        pitches(True, True)
        nv.call("btvae_linear_dgrad_gstats_pw", P(buf(dY, ln)), P(W), None, P(buf(Rk, lk)), P(buf(Yk, lk)), P(mean_k), P(inv_k), P(D3), P(gpart), M, N, K)
        res["gstats"] = (D3, K); res["gstats.gpart"] = (gpart, None)
        dW = torch.zeros(N, K, device=dev)
        pitches(); nv.call("btvae_linear_wgrad", P(buf(dY, ln)), P(buf(X, lk)), lk, P(dW), P(ws), ws.numel(), M, N, K, 0)
        res["wgrad"] = (dW, None)
        dW2, cwp, cwb = torch.zeros(N, K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)
        pitches(); nv.call("btvae_linear_wgrad_affine", P(buf(dY, ln)), P(buf(Yk, lk)), lk, P(sc_k), P(sh_k), P(dbc), P(W), P(cwp), P(cwb), 0,
                           P(dW2), P(ws), ws.numel(), M, N, K, 0)
        res["wgrad_affine"] = (dW2, None); res["wgrad_affine.cols"] = (torch.stack([cwp, cwb]), None)
        scratch = torch.zeros(L.btvae_bn_relu_bwd_scratch(M, N), device=dev)
        dg, db, dbias, dcur = (torch.zeros(N, device=dev) for _ in range(4))
        dP = out(N, ln)
        pitches(); nv.call("btvae_bn_relu_bwd", P(buf(dY, ln)), P(buf(Yn, ln)), P(mean_n), P(inv_n), P(gam_n), P(keep), 1.25, P(dg), P(db), P(dbias), P(dcur),
                           P(dP), P(scratch), M, N, 0)
        res["bn_relu_bwd"] = (dP, N); res["bn_relu_bwd.vec"] = (torch.stack([dg, db, dbias, dcur]), None)
        dg2, db2, sums = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(2, N, device=dev)
        pitches(); nv.call("btvae_bn_relu_bwd_sums", P(buf(dY, ln)), P(buf(Yn, ln)), P(mean_n), P(inv_n), P(keep), 1.25, P(dg2), P(db2), P(sums), P(scratch), M, N, 0)
        res["bwd_sums"] = (torch.cat([sums, torch.stack([dg2, db2])]), None)
        dbias3, dP3 = torch.zeros(N, device=dev), out(N, ln)
        pitches(); nv.call("btvae_bn_relu_bwd_apply", P(buf(dY, ln)), P(buf(Yn, ln)), P(mean_n), P(inv_n), P(gam_n), P(keep), 1.25, P(sums_n), M, P(dbias3), None,
                           P(dP3), P(scratch), M, N, 0)
        res["bwd_apply"] = (dP3, N); res["bwd_apply.dbias"] = (dbias3, None)
        dbias4, dP4 = torch.zeros(N, device=dev), out(N, ln)
        pitches(); nv.call("btvae_relu_bwd", P(buf(dY, ln)), P(buf(Yn, ln)), P(keep), 1.25, P(dbias4), None, P(dP4), P(scratch), M, N, 0)
        res["relu_bwd"] = (dP4, N); res["relu_bwd.dbias"] = (dbias4, None)
        cs = torch.zeros(N, device=dev)
        pitches(); nv.call("btvae_colsum", P(buf(dY, ln)), P(cs), None, P(scratch), M, N, 0)
        res["colsum"] = (cs, None)
        torch.cuda.synchronize()
        return res

    dense, pitched = run(False), run(True)
    for name, (d, cols) in dense.items():
        q = pitched[name][0]
        if cols is None:
            assert torch.equal(d, q), name
        else:
            assert torch.equal(d, q[:, :cols]), name
            assert bool((q[:, cols:] == -3.25).all()), name + ": padding written"
    # the request lasted one launch: a dense call right after a pitched one reads dense rows
    L.btvae_next_row_pitches(LN, 0)
    cs1, cs2, scratch = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(chunks * N, device=dev)
    big = torch.full((M, LN), 7.5, device=dev); big[:, :N] = dY
    nv.call("btvae_colsum", P(big), P(cs1), None, P(scratch), M, N, 0)
    nv.call("btvae_colsum", P(dY), P(cs2), None, P(scratch), M, N, 0)
    assert torch.equal(cs1, cs2)
    L.btvae_next_row_pitches(N - 4, 0)
    with pytest.raises(RuntimeError, match="row pitch"):
        nv.call("btvae_colsum", P(dY), P(cs2), None, P(scratch), M, N, 0)


@pytest.mark.parametrize("M,K,N,ldx,ldr,ldo", [(300, 1000, 1000, 1024, 1024, 1024), (130, 52, 100, 52, 128, 104), (257, 40, 24, 64, 24, 32)])
def test_linear_relu_affine_with_row_pitches(dev, gemm_mode, M, K, N, ldx, ldr, ldo):
    """``btvae_linear_relu_affine_fwd_pw_ld``: activation, residual and output in buffers with row pitches of their own give
    the bits of the dense call, leave the padding alone, and a pitch below N is refused."""
    X, W, b = rnd(M, K, seed=1).float().to(dev), rnd(N, K, seed=2, scale=0.3).float().to(dev), rnd(N, seed=3).float().to(dev)
    sc, sh, R = rnd(N, seed=4).float().to(dev), rnd(N, seed=5).float().to(dev), rnd(M, N, seed=6).float().to(dev)
    dense = torch.empty(M, N, device=dev)
    nv.call("btvae_linear_relu_affine_fwd_pw", P(X), K, P(W), None, P(b), P(sc), P(sh), P(R), P(dense), M, K, N)
    Xp, Rp = torch.full((M, ldx), 7.0, device=dev), torch.full((M, ldr), 7.0, device=dev)
    Xp[:, :K], Rp[:, :N] = X, R
    Op = torch.full((M, ldo), -3.0, device=dev)
    nv.call("btvae_linear_relu_affine_fwd_pw_ld", P(Xp), ldx, P(W), None, P(b), P(sc), P(sh), P(Rp), ldr, P(Op), ldo, M, K, N)
    assert torch.equal(Op[:, :N], dense)
    assert bool((Op[:, N:] == -3.0).all())
    with pytest.raises(RuntimeError, match="row pitch"):
        nv.call("btvae_linear_relu_affine_fwd_pw_ld", P(Xp), ldx, P(W), None, P(b), P(sc), P(sh), P(Rp), ldr, P(Op), N - 4, M, K, N)


@pytest.mark.parametrize("M,K,N", [(200, 96, 100), (300, 1000, 48), (130, 52, 24), (257, 40, 10), (64, 1000, 64)])
def test_linear_relu_affine_and_head(dev, gemm_mode, M, K, N):
    """N <= 64 takes the narrow head kernel in the split mode (N=10: its scalar-store path), N=100 the 128-wide tile."""
    X, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.3), rnd(N, seed=3)
    sc, sh, R = rnd(N, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
    d = [t.float().to(dev) for t in (X, W, b, sc, sh, R)]
    out = torch.empty(M, N, device=dev)
    nv.call("btvae_linear_relu_affine_fwd", P(d[0]), K, P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(d[5]), P(out), M, K, N)
    c = [t.cpu().double() for t in d]
    ref = torch.relu(c[0] @ c[1].t() + c[2]) * c[3] + c[4] + c[5]
    close(out, ref, 1e-5, 2e-6 * math.sqrt(K) + 2e-5, "affine")
    raw = c[0] @ c[1].t() + c[2]
    for act, lo, hi in ((0, -0.5, 0.8), (1, 0.05, 1.0), (2, 1e-4, 0.62)):
        o, df, sq = (torch.empty(M, N, device=dev) for _ in range(3))
        nv.call("btvae_linear_head_fwd", P(d[0]), K, P(d[1]), P(d[2]), act, lo, hi, P(o), P(df), P(sq), M, K, N)
        rr = raw.clone().requires_grad_(True)
        if act == 0:
            val = rr.clamp(lo, hi)
        elif act == 1:
            val = torch.nn.functional.softplus(rr, beta=1, threshold=5).clamp(lo, hi)
        else:
            val = torch.relu(rr).clamp(lo, hi)
        val.sum().backward()
        close(o, val, 1e-5, 2e-6 * math.sqrt(K) + 2e-5, f"head{act}")
        close(sq, val ** 2, 1e-5, 4e-6 * math.sqrt(K) + 2e-5, f"head{act}.sq")
        # the derivative factor may differ where raw sits within round-off of a clamp edge
        edge = ((raw - lo).abs() < 1e-4) | ((raw - hi).abs() < 1e-4) | (raw.abs() < 1e-4)
        if act == 1:
            sp = torch.nn.functional.softplus(raw, beta=1, threshold=5)
            edge = ((sp - lo).abs() < 1e-4) | ((sp - hi).abs() < 1e-4) | ((raw - 5).abs() < 1e-4)
        dd = (df.cpu().double() - rr.grad).abs()
        assert dd[~edge].max() < 1e-5
        # with a workspace the reduction is split over workgroups (K >= 256, few row tiles) and the non-linearity applied
        # where the slabs are summed; without room for two slabs it is the same launch as above
        for ws in (torch.empty(4 * 1024 * 1024, device=dev), torch.empty(M * N + 5, device=dev)):
            o2, df2, sq2 = (torch.full((M, N), 7.0, device=dev) for _ in range(3))
            nv.call("btvae_linear_head_fwd_ws", P(d[0]), K, P(d[1]), P(d[2]), act, lo, hi, P(o2), P(df2), P(sq2), P(ws), ws.numel(), M, K, N)
            close(o2, val, 1e-5, 2e-6 * math.sqrt(K) + 2e-5, f"head{act} split reduction")
            close(sq2, val ** 2, 1e-5, 4e-6 * math.sqrt(K) + 2e-5, f"head{act}.sq split reduction")
            assert (df2.cpu().double() - rr.grad).abs()[~edge].max() < 1e-5
            if ws.numel() < 2 * M * N:
                assert torch.equal(o2, o) and torch.equal(df2, df) and torch.equal(sq2, sq)


@pytest.mark.parametrize("M,N,K", [(300, 1000, 52), (4096, 100, 1000), (130, 48, 1000), (70, 9, 33)])
def test_dgrad_wgrad(dev, gemm_mode, M, N, K):
    dY, W, X, R = rnd(M, N, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(M, K, seed=3), rnd(M, K, seed=4)
    d = [t.float().to(dev) for t in (dY, W, X, R)]
    c = [t.cpu().double() for t in d]
    dX = torch.empty(M, K, device=dev)
    nv.call("btvae_linear_dgrad", P(d[0]), P(d[1]), P(d[3]), P(dX), K, M, N, K)
    close(dX, c[0] @ c[1] + c[3], 2e-6, 3e-6 * math.sqrt(N), "dgrad")
    ws = torch.empty(16 * N * K, device=dev)
    dW = torch.full((N, K), 0.5, device=dev)
    nv.call("btvae_linear_wgrad", P(d[0]), P(d[2]), K, P(dW), P(ws), ws.numel(), M, N, K, 0)
    close(dW, c[0].t() @ c[2], 2e-6, 3e-6 * math.sqrt(M), "wgrad")
    nv.call("btvae_linear_wgrad", P(d[0]), P(d[2]), K, P(dW), P(ws), ws.numel(), M, N, K, 1)
    close(dW, 2 * (c[0].t() @ c[2]), 2e-6, 6e-6 * math.sqrt(M), "wgrad accumulate")


@pytest.mark.parametrize("M,N,K", [(300, 1000, 1000), (130, 48, 1000), (70, 9, 33)])
def test_bn_fold_linear_and_affine_wgrad(dev, gemm_mode, M, N, K):
    """A Linear fed by a train-mode BatchNorm, computed from the BatchNorm's input: folded weights for the forward
    product, scale/shift correction for the weight gradient (accumulating and overwriting)."""
    Y = torch.relu(rnd(M, K, seed=3)); W = rnd(N, K, seed=4) * 0.1; b = rnd(N, seed=5)
    mean = Y.mean(0); inv = 1 / torch.sqrt(Y.var(0, unbiased=False) + 1e-5)
    gam = 1 + 0.3 * rnd(K, seed=6); bet = 0.2 * rnd(K, seed=7)
    dY = rnd(M, N, seed=8)
    sc = inv * gam; sh = bet - mean * sc
    X = Y * sc + sh
    ref_out = X @ W.T + b
    ref_dW = dY.T @ X
    d = {k: v.float().to(dev) for k, v in dict(Y=Y, W=W, b=b, mean=mean, inv=inv, gam=gam, bet=bet, dY=dY).items()}
    Wf, bf = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    scd, shd = torch.empty(K, device=dev), torch.empty(K, device=dev)
    nv.call("btvae_bn_fold_linear", P(d["W"]), P(d["b"]), P(d["mean"]), P(d["inv"]), P(d["gam"]), P(d["bet"]), P(Wf), P(bf), P(scd), P(shd), N, K)
    close(scd, sc, 1e-5, 1e-6, "scale"); close(shd, sh, 1e-5, 1e-6, "shift")
    out = torch.empty(M, N, device=dev)
    nv.call("btvae_matmul_nt_bias", P(d["Y"]), K, P(Wf), K, P(bf), P(out), N, M, N, K)
    close(out, ref_out, 1e-4, 2e-5 * math.sqrt(K), "folded forward")
    ws = torch.empty(8 * 1024 * 1024, device=dev)
    cs = torch.empty(nv.lib().btvae_strip_chunks(M) * N, device=dev)
    dbc = torch.empty(N, device=dev)
    nv.call("btvae_colsum", P(d["dY"]), None, P(dbc), P(cs), M, N, 0)
    dW = torch.full((N, K), 0.5, device=dev)
    nv.call("btvae_linear_wgrad_affine", P(d["dY"]), P(d["Y"]), K, P(scd), P(shd), P(dbc), None, None, None, 0, P(dW), P(ws), ws.numel(), M, N, K, 1)
    close(dW, 0.5 + ref_dW, 1e-4, 2e-5 * math.sqrt(M), "affine wgrad (accumulate)")
    # with the column by-products: sum_m G*Y and sum_m G for G = dY W, and from them the BatchNorm-backward sums
    cwp, cwb = torch.full((K,), 0.25, device=dev), torch.full((K,), -0.5, device=dev)
    nv.call("btvae_linear_wgrad_affine", P(d["dY"]), P(d["Y"]), K, P(scd), P(shd), P(dbc), P(d["W"]), P(cwp), P(cwb), 1, P(dW), P(ws), ws.numel(), M, N, K, 0)
    close(dW, ref_dW, 1e-4, 2e-5 * math.sqrt(M), "affine wgrad")
    G = dY @ W
    s = math.sqrt(M * N)
    close(cwp, 0.25 + (G * Y).sum(0), 1e-4, 1e-5 * s, "sum G*Y")
    close(cwb, -0.5 + G.sum(0), 1e-4, 1e-5 * s, "sum G")
    nv.call("btvae_linear_wgrad_affine", P(d["dY"]), P(d["Y"]), K, P(scd), P(shd), P(dbc), P(d["W"]), P(cwp), P(cwb), 0, P(dW), P(ws), ws.numel(), M, N, K, 0)
    sums, dgam, dbet = torch.empty(2, K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    nv.call("btvae_bn_bwd_sums_from_products", P(cwp), P(cwb), P(d["mean"]), P(d["inv"]), P(dgam), P(dbet), P(sums), K, 0)
    xhat = (Y - mean) * inv
    close(sums[0], G.sum(0), 1e-4, 1e-5 * s, "dbeta from products")
    close(sums[1], (G * xhat).sum(0), 1e-4, 2e-5 * s, "dgamma from products")
    assert torch.equal(dgam, sums[1]) and torch.equal(dbet, sums[0])


@pytest.mark.parametrize("M,N,K,with_r", [(300, 100, 1000, True), (130, 48, 72, False), (1000, 1000, 1000, True)])
def test_dgrad_fused_with_batchnorm_backward(dev, gemm_mode, M, N, K, with_r):
    """dX product fused with the producing unit's BatchNorm+ReLU backward against fp64: dPre and its column sums,
    for column sums (sum G, sum G*xhat) taken from G itself."""
    dY = rnd(M, N, seed=21); W = rnd(N, K, seed=22) * 0.1
    R = rnd(M, K, seed=23) if with_r else None
    Yb = torch.relu(rnd(M, K, seed=24) + 0.2)
    gam = 1 + 0.3 * rnd(K, seed=25)
    mean = Yb.mean(0); inv = 1 / torch.sqrt(Yb.var(0, unbiased=False) + 1e-5)
    G = dY @ W + (R if with_r else 0)
    xhat = (Yb - mean) * inv
    s0, s1 = G.sum(0), (G * xhat).sum(0)
    ref = torch.where(Yb > 0, gam * inv * (G - s0 / M - xhat * s1 / M), torch.zeros_like(G))
    f = lambda t: None if t is None else t.float().to(dev).contiguous()
    d = {k: f(v) for k, v in dict(dY=dY, W=W, R=R, Yb=Yb, gam=gam, mean=mean, inv=inv, sums=torch.stack([s0, s1])).items()}
    tiles = (M + 127) // 128
    dpre, part = torch.empty(M, K, device=dev), torch.empty(tiles, K, device=dev)
    nv.call("btvae_linear_dgrad_bnbwd", P(d["dY"]), P(d["W"]), P(d["R"]), P(d["Yb"]), P(d["mean"]), P(d["inv"]), P(d["gam"]), P(d["sums"]), M,
            P(dpre), P(part), M, N, K)
    scale = float((gam * inv).abs().max())
    close(dpre, ref, 1e-4, scale * (2e-6 * math.sqrt(N) + 1e-6 * math.sqrt(M)), "fused dPre")
    db, cur = torch.ones(K, device=dev), torch.empty(K, device=dev)
    nv.call("btvae_colsum_partials", P(part), tiles, K, P(db), P(cur), 1)
    close(cur, ref.sum(0), 1e-4, scale * 1e-5 * math.sqrt(M) * math.sqrt(N), "dbias")
    close(db, 1 + ref.sum(0), 1e-4, scale * 1e-5 * math.sqrt(M) * math.sqrt(N), "dbias accumulated")


@pytest.mark.parametrize("M,N,K,with_r", [(300, 1000, 1000, True), (130, 72, 48, False)])
def test_dgrad_with_batchnorm_backward_sums(dev, gemm_mode, M, N, K, with_r):
    """dX product that also leaves sum G and sum G*xhat per column (per row tile, then merged) against fp64."""
    dY = rnd(M, N, seed=31); W = rnd(N, K, seed=32) * 0.1
    R = rnd(M, K, seed=33) if with_r else None
    Yb = torch.relu(rnd(M, K, seed=34) + 0.2)
    mean = Yb.mean(0); inv = 1 / torch.sqrt(Yb.var(0, unbiased=False) + 1e-5)
    G = dY @ W + (R if with_r else 0)
    xhat = (Yb - mean) * inv
    f = lambda t: None if t is None else t.float().to(dev).contiguous()
    d = {k: f(v) for k, v in dict(dY=dY, W=W, R=R, Yb=Yb, mean=mean, inv=inv).items()}
    tiles = (M + 127) // 128
    dx, part = torch.empty(M, K, device=dev), torch.empty(tiles, 2, K, device=dev)
    nv.call("btvae_linear_dgrad_gstats", P(d["dY"]), P(d["W"]), P(d["R"]), P(d["Yb"]), P(d["mean"]), P(d["inv"]), P(dx), P(part), M, N, K)
    close(dx, G, 2e-6, 2e-6 * math.sqrt(N), "G")
    sums, dgam, dbet = torch.empty(2, K, device=dev), torch.ones(K, device=dev), torch.ones(K, device=dev)
    nv.call("btvae_bn_bwd_sums_from_partials", P(part), tiles, K, P(dgam), P(dbet), P(sums), 1)
    s = math.sqrt(M) * (1 + float(inv.max()))
    close(sums[0], G.sum(0), 1e-5, 3e-6 * s, "sum G")
    close(sums[1], (G * xhat).sum(0), 1e-5, 3e-6 * s * 3, "sum G*xhat")
    close(dbet, 1 + G.sum(0), 1e-5, 3e-6 * s, "dbeta accumulated")
    close(dgam, 1 + (G * xhat).sum(0), 1e-5, 3e-6 * s * 3, "dgamma accumulated")


@pytest.mark.parametrize("M,N,K", [(300, 1000, 52), (257, 1000, 1000), (130, 72, 48), (1000, 100, 1000), (70, 36, 33)])
def test_presplit_weight_products(dev, gemm_mode, M, N, K):
    """Forward and dX products reading the weight from planes split once (btvae_weight_planes): the planes hold the
    exact three-way truncation split, zero padded; the products are bit-identical to the entry points that split the
    weight tile by tile (same pieces, same order of the same MFMAs), in every mode and for shapes that fall back."""
    X, W, b = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=0.2), rnd(N, seed=43, scale=0.1)
    dY, R = rnd(M, N, seed=44), rnd(M, K, seed=45)
    Xd, Wd, bd, dYd, Rd = (t.float().to(dev) for t in (X, W, b, dY, R))
    n_el = nv.lib().btvae_weight_planes_elems(N, K)
    FT, FK, DT, DK = (N + 127) // 128, (K + 15) // 16, (K + 127) // 128, (N + 15) // 16
    assert n_el == (FT * FK + DT * DK) * 3 * 2048
    planes = torch.full((n_el,), 0x7FC0, dtype=torch.int16, device=dev)
    nv.call("btvae_weight_planes", P(Wd), K, N, K, P(planes), 1)
    assert bool((planes[FT * FK * 3 * 2048:] == 0x7FC0).all()), "which = 1 leaves the dX image alone"
    nv.call("btvae_weight_planes", P(Wd), K, N, K, P(planes), 2)
    # pieces: h = W truncated to bf16, m = (W - h) truncated, l = (W - h - m) truncated; all differences exact in fp32
    Wp = torch.zeros(max(FT * 128, DK * 16), max(FK * 16, DT * 128), device=dev); Wp[:N, :K] = Wd
    trunc = lambda t: (t.view(torch.int32) & -65536).view(torch.float32)
    h = trunc(Wp); m = trunc(Wp - h); l = trunc(Wp - h - m)
    pieces = torch.stack([(p_.view(torch.int32) >> 16).to(torch.int16) for p_ in (h, m, l)])      # [3, rows, cols]
    # forward image: blocks [FT][FK][3][2048]; element (row, k) of a block at the kernel's swizzled LDS offset
    row = torch.arange(128, device=dev)[:, None]; kk = torch.arange(16, device=dev)[None, :]
    pr = (row & ~3) | ((row ^ (row >> 2)) & 3)
    off = pr * 16 + (((kk >> 3) ^ ((pr >> 3) & 1)) << 3) + (kk & 7)                                  # [128, 16]
    imgF = planes[:FT * FK * 3 * 2048].view(FT, FK, 3, 2048)
    wantF = pieces[:, :FT * 128, :FK * 16].reshape(3, FT, 128, FK, 16).permute(1, 3, 0, 2, 4)          # [FT, FK, 3, 128, 16]
    assert torch.equal(imgF[..., off.reshape(-1)].reshape(FT, FK, 3, 128, 16), wantF), "forward image"
    # dX image: blocks [DT][DK][3][16 k rows][128 columns]
    imgD = planes[FT * FK * 3 * 2048:].view(DT, DK, 3, 16, 128)
    wantD = pieces[:, :DK * 16, :DT * 128].reshape(3, DK, 16, DT, 128).permute(3, 1, 0, 2, 4)
    assert torch.equal(imgD, wantD), "dX image"
    tiles = (M + 127) // 128
    Y0, Y1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    p0, p1 = torch.empty(tiles, 2, N, device=dev), torch.empty(tiles, 2, N, device=dev)
    nv.call("btvae_linear_relu_stats_fwd", P(Xd), K, P(Wd), P(bd), P(Y0), P(p0), M, K, N)
    nv.call("btvae_linear_relu_stats_fwd_pw", P(Xd), K, P(Wd), P(planes), P(bd), P(Y1), P(p1), M, K, N)
    assert torch.equal(Y0, Y1) and torch.equal(p0, p1), "relu_stats forward"
    sc, sh = rnd(N, seed=46).float().to(dev), rnd(N, seed=47).float().to(dev)
    Rn = rnd(M, N, seed=48).float().to(dev)
    nv.call("btvae_linear_relu_affine_fwd", P(Xd), K, P(Wd), P(bd), P(sc), P(sh), P(Rn), P(Y0), M, K, N)
    nv.call("btvae_linear_relu_affine_fwd_pw", P(Xd), K, P(Wd), P(planes), P(bd), P(sc), P(sh), P(Rn), P(Y1), M, K, N)
    assert torch.equal(Y0, Y1), "relu_affine forward"
    D0, D1 = torch.empty(M, K, device=dev), torch.empty(M, K, device=dev)
    nv.call("btvae_linear_dgrad", P(dYd), P(Wd), P(Rd), P(D0), K, M, N, K)
    nv.call("btvae_linear_dgrad_pw", P(dYd), P(Wd), P(planes), P(Rd), P(D1), K, M, N, K)
    assert torch.equal(D0, D1), "dgrad"
    close(D1, dYd.cpu().double() @ Wd.cpu().double() + Rd.cpu().double(), 2e-6, 3e-6 * math.sqrt(N), "dgrad vs fp64")
    if K % 4 == 0:
        Yb = torch.relu(rnd(M, K, seed=49) + 0.2).float().to(dev)
        mean, inv = Yb.mean(0), 1 / torch.sqrt(Yb.var(0, unbiased=False) + 1e-5)
        gam, sums = (1 + 0.3 * rnd(K, seed=50)).float().to(dev), rnd(2, K, seed=51).float().to(dev)
        c0, c1 = torch.empty(tiles, K, device=dev), torch.empty(tiles, K, device=dev)
        nv.call("btvae_linear_dgrad_bnbwd", P(dYd), P(Wd), P(Rd), P(Yb), P(mean), P(inv), P(gam), P(sums), M, P(D0), P(c0), M, N, K)
        nv.call("btvae_linear_dgrad_bnbwd_pw", P(dYd), P(Wd), P(planes), P(Rd), P(Yb), P(mean), P(inv), P(gam), P(sums), M, P(D1), P(c1), M, N, K)
        assert torch.equal(D0, D1) and torch.equal(c0, c1), "dgrad + BatchNorm backward"
        g0, g1 = torch.empty(tiles, 2, K, device=dev), torch.empty(tiles, 2, K, device=dev)
        nv.call("btvae_linear_dgrad_gstats", P(dYd), P(Wd), P(Rd), P(Yb), P(mean), P(inv), P(D0), P(g0), M, N, K)
        nv.call("btvae_linear_dgrad_gstats_pw", P(dYd), P(Wd), P(planes), P(Rd), P(Yb), P(mean), P(inv), P(D1), P(g1), M, N, K)
        assert torch.equal(D0, D1) and torch.equal(g0, g1), "dgrad + BatchNorm backward sums"


def absmax_of(t):
    """A maximum slot (two floats: max |t| and the range statistic of the two-piece guard) filled by ``btvae_absmax``."""
    out = torch.zeros(2, device=t.device)
    nv.call("btvae_absmax", P(t), t.numel(), P(out))
    return out


@pytest.mark.parametrize("n", [1, 3, 1000, 4096 * 1000 + 3])
def test_absmax(dev, n):
    x = rnd(n + 1, seed=70).float().to(dev)
    x[n // 2] = -7.5
    for view in (x[:n], x[1:n + 1]):                 # 16-byte aligned and not
        got = absmax_of(view.contiguous() if view.data_ptr() % 16 == 0 else view)
        assert got[0].item() == view.abs().max().item()
    keep = torch.tensor([100.0, 0.0], device=dev)    # extends a maximum already in the slot
    nv.call("btvae_absmax", P(x[:n]), n, P(keep))
    assert keep[0].item() == 100.0
    z = torch.zeros(8, device=dev)
    assert absmax_of(z).tolist() == [0.0, 0.0]


@pytest.fixture
def two_piece():
    nv.set_gemm_precision("fp32_split2")
    yield
    nv.lib().btvae_gemm_operand_maxima(None, None)
    nv.set_gemm_precision("fp32")


@pytest.mark.parametrize("M,N,K,sx,sg", [(300, 1000, 52, 1.0, 1.0), (257, 1000, 1000, 1.0, 1.0), (130, 72, 48, 30.0, 1e-6),
                                         (1000, 100, 1000, 1e-3, 1e4), (640, 1000, 1000, 1.0, 1e-7)])
def test_two_piece_products(dev, two_piece, M, N, K, sx, sg):
    """Linear products on two scaled fp16 pieces per value (given the operands' maxima) against fp64, to the tolerances
    of the fp32 products relative to the operands' scales; a product without maxima is bit-identical to the three-piece
    mode.  Activations and gradients of very different magnitudes exercise the per-tensor scales."""
    X, W, b = rnd(M, K, seed=41, scale=sx), rnd(N, K, seed=42, scale=0.2), rnd(N, seed=43, scale=0.1 * sx)
    dY, R = rnd(M, N, seed=44, scale=sg), rnd(M, K, seed=45, scale=sg)
    Xd, Wd, bd, dYd, Rd = (t.float().to(dev) for t in (X, W, b, dY, R))
    c = {k: v.cpu().double() for k, v in dict(X=Xd, W=Wd, b=bd, dY=dYd, R=Rd).items()}
    mx = {k: absmax_of(v) for k, v in dict(X=Xd, W=Wd, dY=dYd).items()}
    L = nv.lib()
    planes = torch.full((L.btvae_weight_planes_elems(N, K),), 0x7FC0, dtype=torch.int16, device=dev)
    nv.call("btvae_weight_planes_f16", P(Wd), K, N, K, P(planes), 3, P(mx["W"]))
    tiles = (M + 127) // 128
    ref_y = torch.relu(c["X"] @ c["W"].t() + c["b"])
    tol_f = 2e-6 * math.sqrt(K) * sx
    Y3 = torch.empty(M, N, device=dev); p3 = torch.empty(tiles, 2, N, device=dev)
    nv.set_gemm_precision("fp32_split")
    nv.call("btvae_linear_relu_stats_fwd", P(Xd), K, P(Wd), P(bd), P(Y3), P(p3), M, K, N)
    nv.set_gemm_precision("fp32_split2")
    for pl in (None, planes):
        Y = torch.empty(M, N, device=dev); part = torch.empty(tiles, 2, N, device=dev)
        # without maxima: the three-piece kernel on the fp32 weight, whatever the planes hold
        nv.call("btvae_linear_relu_stats_fwd_pw", P(Xd), K, P(Wd), P(pl), P(bd), P(Y), P(part), M, K, N)
        assert torch.equal(Y, Y3) and torch.equal(part, p3)
        L.btvae_gemm_operand_maxima(P(mx["X"]), P(mx["W"]))
        nv.call("btvae_linear_relu_stats_fwd_pw", P(Xd), K, P(Wd), P(pl), P(bd), P(Y), P(part), M, K, N)
        close(Y, ref_y, 2e-6, tol_f, "relu_stats")
        if K >= 1000:
            assert not torch.equal(Y, Y3), "the two-piece kernel did not run"
        mean, invstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
        nv.call("btvae_bn_stats_finalize", P(part), tiles, M, N, P(mean), P(invstd), None, None, 0.1, 1e-5)
        close(mean, Y.cpu().double().mean(0), 1e-5, 1e-6 * sx, "mean")
        # the maxima were consumed: the next product is a three-piece one again
        nv.call("btvae_linear_relu_stats_fwd_pw", P(Xd), K, P(Wd), P(pl), P(bd), P(Y), P(part), M, K, N)
        assert torch.equal(Y, Y3)
        sc, sh, Rn = rnd(N, seed=46).float().to(dev), rnd(N, seed=47).float().to(dev), rnd(M, N, seed=48).float().to(dev)
        L.btvae_gemm_operand_maxima(P(mx["X"]), P(mx["W"]))
        nv.call("btvae_linear_relu_affine_fwd_pw", P(Xd), K, P(Wd), P(pl), P(bd), P(sc), P(sh), P(Rn), P(Y), M, K, N)
        close(Y, ref_y * sc.cpu().double() + sh.cpu().double() + Rn.cpu().double(), 1e-5, tol_f * 3 + 2e-5, "relu_affine")
        D = torch.empty(M, K, device=dev)
        ref_g = c["dY"] @ c["W"] + c["R"]
        tol_d = 3e-6 * math.sqrt(N) * sg
        L.btvae_gemm_operand_maxima(P(mx["dY"]), P(mx["W"]))
        nv.call("btvae_linear_dgrad_pw", P(dYd), P(Wd), P(pl), P(Rd), P(D), K, M, N, K)
        close(D, ref_g, 2e-6, tol_d, "dgrad")
        if K % 4 == 0:
            Yb = torch.relu(rnd(M, K, seed=49) + 0.2).float().to(dev)
            mean, inv = Yb.mean(0), 1 / torch.sqrt(Yb.var(0, unbiased=False) + 1e-5)
            gam = (1 + 0.3 * rnd(K, seed=50)).float().to(dev)
            xhat = (Yb.cpu().double() - mean.cpu().double()) * inv.cpu().double()
            g_part = torch.empty(tiles, 2, K, device=dev)
            L.btvae_gemm_operand_maxima(P(mx["dY"]), P(mx["W"]))
            nv.call("btvae_linear_dgrad_gstats_pw", P(dYd), P(Wd), P(pl), P(Rd), P(Yb), P(mean), P(inv), P(D), P(g_part), M, N, K)
            close(D, ref_g, 2e-6, tol_d, "dgrad + sums: G")
            sums, dgam, dbet = torch.empty(2, K, device=dev), torch.zeros(K, device=dev), torch.zeros(K, device=dev)
            nv.call("btvae_bn_bwd_sums_from_partials", P(g_part), tiles, K, P(dgam), P(dbet), P(sums), 0)
            s = math.sqrt(M) * (1 + float(inv.max())) * sg * math.sqrt(N) * 0.2
            close(sums[0], ref_g.sum(0), 1e-5, 3e-6 * s, "sum G")
            close(sums[1], (ref_g * xhat).sum(0), 1e-5, 9e-6 * s, "sum G*xhat")
            s0, s1 = ref_g.sum(0), (ref_g * xhat).sum(0)
            ref_dpre = torch.where(Yb.cpu() > 0, gam.cpu().double() * inv.cpu().double() * (ref_g - s0 / M - xhat * s1 / M), torch.zeros_like(ref_g))
            sums_d = torch.stack([s0, s1]).float().to(dev)
            cp = torch.empty(tiles, K, device=dev)
            L.btvae_gemm_operand_maxima(P(mx["dY"]), P(mx["W"]))
            nv.call("btvae_linear_dgrad_bnbwd_pw", P(dYd), P(Wd), P(pl), P(Rd), P(Yb), P(mean), P(inv), P(gam), P(sums_d), M, P(D), P(cp), M, N, K)
            scale = float((gam * inv).abs().max()) * sg
            close(D, ref_dpre, 1e-4, scale * (2e-6 * math.sqrt(N) + 1e-6 * math.sqrt(M)), "dgrad + BatchNorm backward")
    ws = torch.empty(16 * N * K, device=dev)
    dW = torch.full((N, K), 0.5, device=dev)
    L.btvae_gemm_operand_maxima(P(mx["dY"]), P(mx["X"]))
    nv.call("btvae_linear_wgrad", P(dYd), P(Xd), K, P(dW), P(ws), ws.numel(), M, N, K, 0)
    close(dW, c["dY"].t() @ c["X"], 2e-6, 3e-6 * math.sqrt(M) * sx * sg, "wgrad")


def test_two_piece_outlier_and_degenerate_scales(dev, two_piece):
    """One element 2^20 above the rest of its tensor: the small elements keep an ABSOLUTE error of ~2^-40 of the maximum
    (the result stays accurate relative to the outlier row's scale); all-zero operands and a maximum of 0 are fine."""
    M, K, N = 256, 1000, 256
    X, W = rnd(M, K, seed=80), rnd(N, K, seed=81, scale=0.2)
    X[3, 5] = 2.0 ** 20
    Xd, Wd, bd = X.float().to(dev), W.float().to(dev), torch.zeros(N, device=dev)
    tiles = (M + 127) // 128
    Y, part = torch.empty(M, N, device=dev), torch.empty(tiles, 2, N, device=dev)
    L = nv.lib()
    mx, mw = absmax_of(Xd), absmax_of(Wd)
    L.btvae_gemm_operand_maxima(P(mx), P(mw))
    nv.call("btvae_linear_relu_stats_fwd", P(Xd), K, P(Wd), P(bd), P(Y), P(part), M, K, N)
    ref = torch.relu(Xd.cpu().double() @ Wd.cpu().double().t())
    err = (Y.cpu().double() - ref).abs()
    assert err[3].max() <= 2e-6 * 2.0 ** 20 * 0.2 * 4                # the outlier's row: relative to its own products
    others = torch.ones(M, dtype=torch.bool); others[3] = False
    # other rows: elements ~1 are ~2^-20 of the maximum, below the second piece's full-precision range (2^-18 of it):
    # their error is the fp16 denormal spacing, 2^-39 of the maximum = 2^-19 absolute per element here
    assert err[others].max() <= 2.0 ** -19 * 0.2 * 4 * math.sqrt(K)
    Z = torch.zeros(M, K, device=dev)
    mz = absmax_of(Z)
    L.btvae_gemm_operand_maxima(P(mz), P(mw))
    nv.call("btvae_linear_relu_stats_fwd", P(Z), K, P(Wd), P(bd), P(Y), P(part), M, K, N)
    assert float(Y.abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(300, 1000, 52), (130, 72, 48), (257, 33, 40)])
def test_output_maximum_from_producers(dev, gemm_mode, M, N, K):
    """btvae_output_maximum: the next producer launch leaves max |what it wrote| in the slot (exactly: a maximum involves
    no rounding), extends a value already there, and the request does not outlive that launch."""
    L = nv.lib()
    X, W, b = rnd(M, K, seed=91).float().to(dev), rnd(N, K, seed=92, scale=0.2).float().to(dev), rnd(N, seed=93, scale=0.1).float().to(dev)
    tiles = (M + 127) // 128
    Y, part = torch.empty(M, N, device=dev), torch.empty(tiles, 2, N, device=dev)
    slot = torch.zeros(2, device=dev)
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_linear_relu_stats_fwd", P(X), K, P(W), P(b), P(Y), P(part), M, K, N)
    assert slot[0].item() == Y.abs().max().item() > 0
    slot[0] = 1e6
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_linear_relu_stats_fwd", P(X), K, P(W), P(b), P(Y), P(part), M, K, N)
    assert slot[0].item() == 1e6
    nv.call("btvae_linear_relu_stats_fwd", P(X), K, P(W), P(b), P(Y), P(part), M, K, N)      # no request pending
    assert slot[0].item() == 1e6
    sc, sh, R = rnd(N, seed=94).float().to(dev), rnd(N, seed=95).float().to(dev), rnd(M, N, seed=96).float().to(dev)
    out = torch.empty(M, N, device=dev)
    slot.zero_()
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_linear_relu_affine_fwd", P(X), K, P(W), P(b), P(sc), P(sh), P(R), P(out), M, K, N)
    assert slot[0].item() == out.abs().max().item()
    mean, inv = Y.mean(0), 1 / torch.sqrt(Y.var(0, unbiased=False) + 1e-5)
    gam, bet = (1 + 0.3 * rnd(N, seed=97)).float().to(dev), rnd(N, seed=98).float().to(dev)
    slot.zero_()
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_bn_apply_fwd", P(Y), P(mean), P(inv), P(gam), P(bet), P(R), None, 1.0, P(out), M, N)
    assert slot[0].item() == out.abs().max().item()
    G = rnd(M, N, seed=99).float().to(dev)
    dpre = torch.empty(M, N, device=dev)
    scr = torch.empty(L.btvae_bn_relu_bwd_scratch(M, N), device=dev)
    dg, dbt, dbi = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    slot.zero_()
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_bn_relu_bwd", P(G), P(Y), P(mean), P(inv), P(gam), None, 1.0, P(dg), P(dbt), P(dbi), None, P(dpre), P(scr), M, N, 0)
    assert slot[0].item() == dpre.abs().max().item()
    slot.zero_()
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_relu_bwd", P(G), P(Y), None, 1.0, P(dbi), None, P(dpre), P(scr), M, N, 0)
    assert slot[0].item() == dpre.abs().max().item()
    if N % 4 == 0:
        dY = rnd(M, K, seed=100).float().to(dev)          # dX product of a Linear [K out, N in]: G = dY Wt, Wt [K, N]
        Wt = rnd(K, N, seed=101, scale=0.2).float().to(dev)
        sums = rnd(2, N, seed=102).float().to(dev)
        cp = torch.empty(tiles, N, device=dev)
        slot.zero_()
        L.btvae_output_maximum(P(slot))
        nv.call("btvae_linear_dgrad_bnbwd", P(dY), P(Wt), None, P(Y), P(mean), P(inv), P(gam), P(sums), M, P(dpre), P(cp), M, K, N)
        assert slot[0].item() == dpre.abs().max().item()
    # folded weights and the decoder's z|c rows
    Wf, bf2, scd, shd = torch.empty(N, K, device=dev), torch.empty(N, device=dev), torch.empty(K, device=dev), torch.empty(K, device=dev)
    mk, ik = X.mean(0), 1 / torch.sqrt(X.var(0, unbiased=False) + 1e-5)
    gk, bk = (1 + 0.3 * rnd(K, seed=103)).float().to(dev), rnd(K, seed=104).float().to(dev)
    slot.zero_()
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_bn_fold_linear", P(W), P(b), P(mk), P(ik), P(gk), P(bk), P(Wf), P(bf2), P(scd), P(shd), N, K)
    assert slot[0].item() == Wf.abs().max().item() > 0
    R_, S_, Ld, Cd = 37, 3, 10, 4
    mu, sg = rnd(R_, Ld, seed=105).float().to(dev), (0.1 + rnd(R_, Ld, seed=106).abs()).float().to(dev)
    ep, cd = rnd(S_ * R_, Ld, seed=107).float().to(dev), (5 * rnd(R_, Cd, seed=108)).float().to(dev)
    zc, kl = torch.empty(S_ * R_, Ld + Cd, device=dev), torch.empty(R_, device=dev)
    slot.zero_()
    L.btvae_output_maximum(P(slot))
    nv.call("btvae_reparam_kl_fwd", P(mu), P(sg), P(ep), None, P(cd), P(zc), P(kl), R_, S_, Ld, Cd, R_, 0, Cd, 0, 0)
    assert slot[0].item() == zc.abs().max().item() > 0
    # a product that is no producer drops the request
    L.btvae_output_maximum(P(slot))
    D = torch.empty(M, K, device=dev)
    nv.call("btvae_linear_dgrad", P(Y), P(W), None, P(D), K, M, N, K)
    before = slot[0].item()
    nv.call("btvae_linear_relu_stats_fwd", P(X), K, P(W), P(b), P(Y), P(part), M, K, N)
    assert slot[0].item() == before


def test_weight_planes_batch(dev):
    """Planes of several matrices in one launch equal the one-matrix calls, image selection included."""
    import ctypes
    shapes = [(1000, 52), (1000, 1000), (48, 1000), (100, 72), (36, 33)] * 4          # 20 entries: two launches of <= 16
    which = [(i % 3) + 1 for i in range(len(shapes))]
    Ws = [rnd(n, k, seed=60 + i).float().to(dev) for i, (n, k) in enumerate(shapes)]
    L = nv.lib()
    sizes = [L.btvae_weight_planes_elems(n, k) for n, k in shapes]
    one = [torch.full((s,), 0x7FC0, dtype=torch.int16, device=dev) for s in sizes]
    many = [torch.full((s,), 0x7FC0, dtype=torch.int16, device=dev) for s in sizes]
    for W, (n, k), pl, wh in zip(Ws, shapes, one, which):
        nv.call("btvae_weight_planes", P(W), k, n, k, P(pl), wh)
    cnt = len(shapes)
    PA, IA = ctypes.c_void_p * cnt, ctypes.c_int * cnt
    nv.call("btvae_weight_planes_multi", PA(*[W.data_ptr() for W in Ws]), IA(*[n for n, _ in shapes]), IA(*[k for _, k in shapes]),
            PA(*[pl.data_ptr() for pl in many]), IA(*which), cnt)
    for i, (a, b) in enumerate(zip(one, many)):
        assert torch.equal(a, b), f"entry {i}"


def test_weight_planes_f16_batch(dev):
    """Two-piece planes of several matrices in one call: each matrix's maximum lands in its slot (neighbouring and
    scattered slots, more than one launch's worth of entries) and the planes equal the one-matrix calls."""
    import ctypes
    shapes = [(1000, 52), (1000, 1000), (48, 1000), (100, 72), (36, 33)] * 4
    which = [(i % 3) + 1 for i in range(len(shapes))]
    Ws = [rnd(n, k, seed=160 + i, scale=0.1 * (i + 1)).float().to(dev) for i, (n, k) in enumerate(shapes)]
    L = nv.lib()
    sizes = [L.btvae_weight_planes_elems(n, k) for n, k in shapes]
    one = [torch.full((s,), 0x7FC0, dtype=torch.int16, device=dev) for s in sizes]
    many = [torch.full((s,), 0x7FC0, dtype=torch.int16, device=dev) for s in sizes]
    slots = torch.full((128,), 123.0, device=dev)                               # 64 two-float slots
    where = list(range(12)) + [20, 40, 41, 42, 50, 13, 14, 63]                  # runs of neighbours and loners
    for W, (n, k), pl, wh in zip(Ws, shapes, one, which):
        nv.call("btvae_weight_planes_f16", P(W), k, n, k, P(pl), wh, P(absmax_of(W)))
    cnt = len(shapes)
    PA, IA = ctypes.c_void_p * cnt, ctypes.c_int * cnt
    nv.call("btvae_weight_planes_f16_multi", PA(*[W.data_ptr() for W in Ws]), IA(*[n for n, _ in shapes]), IA(*[k for _, k in shapes]),
            PA(*[pl.data_ptr() for pl in many]), IA(*which), PA(*[slots[2 * w:2 * w + 2].data_ptr() for w in where]), cnt)
    for i, (a, b) in enumerate(zip(one, many)):
        assert torch.equal(a, b), f"entry {i}"
        assert slots[2 * where[i]].item() == Ws[i].abs().max().item(), f"maximum {i}"
    untouched = [j for i in range(64) if i not in where for j in (2 * i, 2 * i + 1)]
    assert bool((slots[untouched] == 123.0).all())


def test_weight_maxima_retaken_by_every_graph_replay(dev):
    """``btvae_weight_planes_f16_multi`` captured into a HIP graph: every replay clears the maximum slots (with a kernel:
    a captured memset does not reliably run in its place) and takes the maxima of the weights as they are then --
    smaller ones included -- and ``btvae_slots_zero`` clears a run of slots."""
    import ctypes
    shapes = [(1000, 52), (1000, 1000), (48, 1000)]
    Ws = [rnd(n, k, seed=170 + i).float().to(dev) for i, (n, k) in enumerate(shapes)]
    L = nv.lib()
    planes = [torch.zeros(L.btvae_weight_planes_elems(n, k), dtype=torch.int16, device=dev) for n, k in shapes]
    slots = torch.full((16,), 5.0e4, device=dev)                                # eight two-float slots
    cnt = len(shapes)
    PA, IA = ctypes.c_void_p * cnt, ctypes.c_int * cnt
    args = (PA(*[W.data_ptr() for W in Ws]), IA(*[n for n, _ in shapes]), IA(*[k for _, k in shapes]),
            PA(*[pl.data_ptr() for pl in planes]), IA(3, 3, 2), PA(slots[2:4].data_ptr(), slots[4:6].data_ptr(), slots[12:14].data_ptr()), cnt)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nv.call("btvae_weight_planes_f16_multi", *args)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        nv.call("btvae_weight_planes_f16_multi", *args)
    for factor in (1.0, 1 / 64, 512.0, 1 / 4096, 3.0):
        for W in Ws:
            W.mul_(factor)
        graph.replay()
        got = slots.cpu()
        for i, j in enumerate((1, 2, 6)):
            assert got[2 * j].item() == Ws[i].abs().max().item(), (factor, i)
        assert bool((got[[0, 1, 6, 7, 8, 9, 10, 11, 14, 15]] == 5.0e4).all())
        one = torch.zeros_like(planes[1])
        nv.call("btvae_weight_planes_f16", P(Ws[1]), 1000, 1000, 1000, P(one), 3, P(slots[4:6]))
        assert torch.equal(one, planes[1]), factor
    nv.call("btvae_slots_zero", P(slots[4:]), 5)
    assert slots.cpu().tolist()[4:9] == [0.0] * 5 and slots[2].item() > 0 and slots[9].item() == 5.0e4


def test_two_piece_guard_flags(dev, two_piece):
    """The guard of the two-piece products: a product launched for a site records, in the host-visible flag array,
    (a) an operand with a region more than 2^16 (the default threshold) below its tensor's maximum -- a planted outlier, or a recorded maximum
    far above the data -- and (b) non-finite accumulators, which is what a recorded maximum below the data produces
    (fp16 overflow; the ReLU epilogue would turn the NaNs into zeros without a trace).  Clean operands raise nothing."""
    import numpy as np
    L = nv.lib()
    flags = np.ctypeslib.as_array(L.btvae_guard_flags(), shape=(2 * L.btvae_guard_sites(),))
    M, K, N = 512, 1000, 256
    X, W = rnd(M, K, seed=180).float().to(dev), rnd(N, K, seed=181, scale=0.2).float().to(dev)
    bd = torch.zeros(N, device=dev)
    Y, part = torch.empty(M, N, device=dev), torch.empty((M + 127) // 128, 2, N, device=dev)
    ref = torch.relu(X.double() @ W.double().t())

    def product(xslot, wslot, site, x=X):
        flags[2 * site:2 * site + 2] = 0
        L.btvae_gemm_operand_maxima(P(xslot), P(wslot))
        L.btvae_gemm_guard_site(site)
        nv.call("btvae_linear_relu_stats_fwd", P(x), K, P(W), P(bd), P(Y), P(part), M, K, N)
        torch.cuda.synchronize()
        return int(flags[2 * site]), int(flags[2 * site + 1])

    mw = absmax_of(W)
    assert product(absmax_of(X), mw, 7) == (0, 0)
    assert float((Y.double() - ref).abs().max()) <= 1e-4
    Xo = X.clone()
    Xo[300, 17] = 2.0 ** 20                                   # one element far above everything else
    assert product(absmax_of(Xo), mw, 8, Xo) == (1, 0)
    big = absmax_of(X)
    big[0] *= 2.0 ** 18                                       # a recorded maximum far above the data (a stale slot)
    assert product(big, mw, 9) == (1, 0)
    small = absmax_of(X)
    small[0] *= 2.0 ** -9                                     # ... and one far below it: fp16 overflow
    assert product(small, mw, 10) == (0, 1)
    assert flags[14] == 0 and flags[15] == 0                  # untouched site
    # the launch-of-its-own form of the range test
    flags[22:24] = 0
    nv.call("btvae_guard_check", P(absmax_of(Xo)), P(mw), 11)
    torch.cuda.synchronize()
    assert flags[22] == 1
    nv.call("btvae_guard_check", P(absmax_of(X)), P(mw), 12)
    torch.cuda.synchronize()
    assert flags[24] == 0
    flags[:] = 0


def test_matmul_bias_and_mul(dev):
    M, K, N = 260, 100, 1176
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    d = [t.float().to(dev) for t in (A, B, bias)]
    C = torch.empty(M, N, device=dev)
    nv.call("btvae_matmul_nt_bias", P(d[0]), K, P(d[1]), K, P(d[2]), P(C), N, M, N, K)
    close(C, d[0].cpu().double() @ d[1].cpu().double().t() + d[2].cpu().double(), 2e-6, 3e-5, "nt_bias")
    E1, E2 = rnd(M, K, seed=4).float().to(dev), rnd(M, K, seed=5).float().to(dev)
    C2 = torch.empty(M, K, device=dev)
    nv.call("btvae_matmul_nn_mul", P(C), N, P(d[1]), K, P(E1), P(E2), K, P(C2), K, M, K, N)
    ref = (C.cpu().double() @ d[1].cpu().double()) * E1.cpu().double() * E2.cpu().double()
    close(C2, ref, 1e-5, 1e-3, "nn_mul")


@pytest.mark.parametrize("M,N", [(1000, 1000), (77, 36), (300, 30)])
def test_batchnorm_fwd_bwd(dev, M, N):
    Yr = torch.relu(rnd(M, N, seed=1)).float()
    G, R = rnd(M, N, seed=2).float(), rnd(M, N, seed=3).float()
    gam, bet = (rnd(N, seed=4) * 0.5 + 1).float(), rnd(N, seed=5).float()
    y = Yr.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    mean = y.mean(0); var = y.var(0, unbiased=False)
    out = (y - mean) / torch.sqrt(var + 1e-5) * g64 + b64 + R.double()
    # forward apply
    d = {k: v.to(dev) for k, v in dict(Y=Yr, G=G, R=R, gam=gam, bet=bet).items()}
    mean_d = mean.detach().float().to(dev); inv_d = (1 / torch.sqrt(var.detach() + 1e-5)).float().to(dev)
    o = torch.empty(M, N, device=dev)
    nv.call("btvae_bn_apply_fwd", P(d["Y"]), P(mean_d), P(inv_d), P(d["gam"]), P(d["bet"]), P(d["R"]), None, 1.0, P(o), M, N)
    close(o, out, 1e-5, 1e-5, "bn fwd")
    # backward: reference goes through relu as well (y = relu(pre))
    pre = rnd(M, N, seed=1).requires_grad_(True)
    y2 = torch.relu(pre)
    m2 = y2.mean(0); v2 = y2.var(0, unbiased=False)
    out2 = (y2 - m2) / torch.sqrt(v2 + 1e-5) * g64 + b64
    out2.backward(G.double())
    nscr = nv.lib().btvae_bn_relu_bwd_scratch(M, N)
    scr = torch.empty(nscr, device=dev)
    dgam, dbet, dbias = (torch.zeros(N, device=dev) for _ in range(3))
    dbcur = torch.empty(N, device=dev)
    dpre = torch.empty(M, N, device=dev)
    nv.call("btvae_bn_relu_bwd", P(d["G"]), P(d["Y"]), P(mean_d), P(inv_d), P(d["gam"]), None, 1.0, P(dgam), P(dbet), P(dbias), P(dbcur), P(dpre), P(scr), M, N, 0)
    s = math.sqrt(M)
    close(dpre, pre.grad, 1e-4, 2e-6, "dpre")
    close(dgam, g64.grad, 1e-5, 3e-6 * s, "dgamma")
    close(dbet, b64.grad, 1e-5, 3e-6 * s, "dbeta")
    close(dbias, pre.grad.sum(0), 1e-5, 3e-6 * s, "dbias")
    assert torch.equal(dbcur, dbias)          # not accumulating: the current sums are the gradient
    cs = torch.empty(nv.lib().btvae_strip_chunks(M) * N, device=dev)
    outc = torch.ones(N, device=dev)
    curc = torch.empty(N, device=dev)
    nv.call("btvae_colsum", P(d["G"]), P(outc), P(curc), P(cs), M, N, 1)
    close(outc, 1 + G.double().sum(0), 1e-5, 3e-6 * s, "colsum")
    close(curc, G.double().sum(0), 1e-5, 3e-6 * s, "colsum current")


@pytest.mark.parametrize("threeD", [False, True])
def test_reparam_kl(dev, threeD):
    B, S, L, C = 37, 5, 48, 4
    R = S * B if threeD else B
    Sl = 1 if threeD else S
    mu, sg = rnd(R, L, seed=1).float(), (rnd(R, L, seed=2).abs() * 0.5 + 0.05).float()
    eps = rnd(Sl * R, L, seed=3).float()
    cond = rnd(B, S, C, seed=4).float() if threeD else rnd(B, C, seed=4).float()
    d = [t.to(dev) for t in (mu, sg, eps, cond)]
    zc = torch.empty(Sl * R, L + C, device=dev); kl = torch.empty(R, device=dev)
    if threeD:
        cdiv, hi, lo = B, C, S * C
    else:
        cdiv, hi, lo = R, 0, C
    nv.call("btvae_reparam_kl_fwd", P(d[0]), P(d[1]), P(d[2]), None, P(d[3]), P(zc), P(kl), R, Sl, L, C, cdiv, hi, lo, 0, 0)
    z_ref = (mu.double() + sg.double() * eps.double().view(Sl, R, L)).view(Sl * R, L)
    if threeD:
        c_ref = cond.double().transpose(0, 1).reshape(R, C)
    else:
        c_ref = cond.double().unsqueeze(0).expand(S, B, C).reshape(S * B, C)
    close(zc[:, :L], z_ref, 1e-6, 1e-6, "z")
    close(zc[:, L:], c_ref, 0, 0, "cond copy")
    kl_ref = (0.5 * (sg.double() ** 2 + mu.double() ** 2 - 1 - torch.log(sg.double() ** 2))).sum(-1)
    close(kl, kl_ref, 1e-5, 1e-5, "kl")
    # backward
    dzc = rnd(Sl * R, L + C, seed=5).float().to(dev)
    dfm, dfs = (rnd(R, L, seed=6) > 0).float().to(dev), torch.rand(R, L, generator=torch.Generator().manual_seed(7)).to(dev)
    dm, ds = torch.empty(R, L, device=dev), torch.empty(R, L, device=dev)
    w = 0.7 / R
    nv.call("btvae_reparam_kl_bwd", P(dzc), P(d[2]), P(d[0]), P(d[1]), P(dfm), P(dfs), P(dm), P(ds), R, Sl, L, C, w)
    dz = dzc.cpu().double()[:, :L].view(Sl, R, L)
    close(dm, (dz.sum(0) + w * mu.double()) * dfm.cpu().double(), 1e-5, 1e-6, "dmu")
    close(ds, ((dz * eps.double().view(Sl, R, L)).sum(0) + w * (sg.double() - 1 / sg.double())) * dfs.cpu().double(), 1e-5, 1e-5, "dsigma")
    # in-kernel Philox draws: standard normal moments, and reproducible
    e1, e2 = torch.empty(Sl * R, L, device=dev), torch.empty(Sl * R, L, device=dev)
    nv.call("btvae_reparam_kl_fwd", P(d[0]), P(d[1]), None, P(e1), P(d[3]), P(zc), P(kl), R, Sl, L, C, cdiv, hi, lo, 2112, 7)
    nv.call("btvae_reparam_kl_fwd", P(d[0]), P(d[1]), None, P(e2), P(d[3]), P(zc), P(kl), R, Sl, L, C, cdiv, hi, lo, 2112, 7)
    assert torch.equal(e1, e2)
    assert abs(e1.mean().item()) < 0.05 and abs(e1.std().item() - 1) < 0.05


@pytest.mark.parametrize("D,V,rows,B", [(48, 100, 301, 43), (6, 9, 50, 10), (24, 100, 130, 130), (64, 70, 37, 37), (2, 3, 5, 1), (33, 40, 4, 4),
                                         (12, 20, 70, 7), (30, 35, 66, 66), (65, 30, 21, 7), (96, 120, 70, 35), (128, 60, 9, 9)])
def test_dictgauss_mvn(dev, D, V, rows, B):
    """Dimensions up to 64 run on the register kernels, beyond that (65, 96, 128) on the one-workgroup-per-row LDS kernels."""
    U = rnd(D, V, seed=1)
    sig = (rnd(rows, V, seed=2).abs() * 0.3 + 1e-3).clamp(max=0.62)
    mu, x = rnd(rows, D, seed=3), rnd(B, D, seed=4)
    m = 0.1
    NP = D * (D + 1) // 2
    Ud = U.float().to(dev)
    Uh, nr, UU, dg = torch.empty(D, V, device=dev), torch.empty(V, device=dev), torch.empty(NP, V, device=dev), torch.empty(NP, device=dev)
    nv.call("btvae_dict_prepare", P(Ud), P(Uh), P(nr), P(UU), P(dg), D, V, m)
    # fp64 reference with autograd
    U64 = Ud.cpu().double().requires_grad_(True)
    s64 = sig.float().double().requires_grad_(True)
    mu64 = mu.float().double().requires_grad_(True)
    x64 = x.float().double()
    Uh64 = U64 / torch.linalg.norm(U64, dim=0, keepdim=True)
    Sg = Uh64 * s64[:, None, :]
    Sigma = Sg @ Sg.transpose(-1, -2) + torch.eye(D, dtype=torch.float64) * m * m
    xr = x64[torch.arange(rows) % B]
    lp = torch.distributions.MultivariateNormal(mu64, covariance_matrix=Sigma).log_prob(xr)
    g = -1.0 / rows
    (g * lp.sum()).backward()
    close(Uh, Uh64, 1e-6, 1e-7, "Uhat")
    # covariance build
    s2 = (sig.float() ** 2).to(dev)
    Sp = torch.empty(rows, NP, device=dev)
    nv.call("btvae_matmul_nt_bias", P(s2), V, P(UU), V, P(dg), P(Sp), NP, rows, NP, V)
    tri = torch.tril_indices(D, D)
    close(Sp, Sigma.detach()[:, tri[0], tri[1]], 1e-5, 1e-6, "Sigma packed")
    mud, xd = mu.float().to(dev), x.float().to(dev)
    lpf = torch.empty(rows, device=dev)
    nv.call("btvae_mvn_rows_fwd", P(Sp), P(mud), P(xd), P(lpf), rows, B, D)
    close(lpf, lp, 2e-5, 1e-4, "logp fwd-only")
    dfm = torch.ones(rows, D, device=dev)
    lpb, dmu, Pg = torch.empty(rows, device=dev), torch.empty(rows, D, device=dev), torch.empty(rows, NP, device=dev)
    nv.call("btvae_mvn_rows_fwd_bwd", P(Sp), P(mud), P(xd), P(dfm), None, g, P(lpb), P(dmu), P(Pg), rows, B, D)
    close(lpb, lp, 2e-5, 1e-4, "logp")
    gs = mu64.grad.abs().max().item()
    close(dmu, mu64.grad, 1e-4, 2e-5 * gs, "dmu")
    # dsigma = (Pg @ UU) * sigma * dfac
    sd_, ones = sig.float().to(dev), torch.ones(rows, V, device=dev)
    dsg = torch.empty(rows, V, device=dev)
    nv.call("btvae_matmul_nn_mul", P(Pg), NP, P(UU), V, P(sd_), P(ones), V, P(dsg), V, rows, V, NP)
    close(dsg, s64.grad, 2e-4, 3e-5 * s64.grad.abs().max().item(), "dsigma")
    # dU via M1 = Pg^T sigma^2
    ws = torch.empty(8 * NP * V, device=dev)
    M1 = torch.empty(NP, V, device=dev)
    nv.call("btvae_matmul_tn", P(Pg), NP, P(s2), V, P(M1), P(ws), ws.numel(), NP, V, rows, 0, 3)
    dU = torch.zeros(D, V, device=dev)
    nv.call("btvae_dict_grad", P(M1), P(Uh), P(nr), P(dU), D, V, 0)
    close(dU, U64.grad, 2e-4, 3e-5 * U64.grad.abs().max().item(), "dU")


def test_diag_loglik(dev):
    rows, B, D = 90, 30, 48
    mu, sg, x = rnd(rows, D, seed=1), rnd(rows, D, seed=2).abs() * 0.5 + 0.1, rnd(B, D, seed=3)
    m64, s64 = mu.float().double().requires_grad_(True), sg.float().double().requires_grad_(True)
    xr = x.float().double()[torch.arange(rows) % B]
    lp = (-0.5 * math.log(2 * math.pi) - torch.log(s64) - 0.5 * ((xr - m64) / s64) ** 2).sum(-1)
    g = -1.0 / rows
    (g * lp.sum()).backward()
    d = [t.float().to(dev) for t in (x, mu, sg)]
    ones = torch.ones(rows, D, device=dev)
    o, dm, ds = torch.empty(rows, device=dev), torch.empty(rows, D, device=dev), torch.empty(rows, D, device=dev)
    nv.call("btvae_diag_gauss_loglik", P(d[0]), P(d[1]), P(d[2]), P(o), P(ones), P(ones), P(dm), P(ds), None, g, rows, B, D)
    close(o, lp, 1e-5, 1e-4, "logp")
    close(dm, m64.grad, 1e-4, 1e-6, "dmu")
    close(ds, s64.grad, 1e-4, 1e-5, "dsigma")


def test_elbo_scalars_and_iw_lse(dev):
    lp, kl = rnd(5000, seed=1).float() * 30 - 100, rnd(300, seed=2).abs().float() * 10
    out = torch.empty(4, device=dev)
    lpd, kld = lp.to(dev), kl.to(dev)
    nv.call("btvae_elbo_scalars", P(lpd), 5000, P(kld), 300, 0.7, P(out))
    rll, k = lp.double().mean(), kl.double().mean()
    close(out, torch.stack([-(rll - 0.7 * k), rll - k, rll, k]), 1e-6, 1e-6, "elbo scalars")
    K, N, L = 11, 50, 24
    mu, sg = rnd(N, L, seed=3).float(), (rnd(N, L, seed=4).abs() * 0.5 + 0.05).float()
    eps, rl = rnd(K, N, L, seed=5).float(), (rnd(K, N, seed=6) * 20 - 50).float()
    z = mu.double() + sg.double() * eps.double()
    nlp = lambda t, m, s: (-0.5 * math.log(2 * math.pi) - torch.log(s) - 0.5 * ((t - m) / s) ** 2).sum(-1)
    w = -nlp(z, mu.double(), sg.double()) + rl.double() + nlp(z, torch.zeros_like(z), torch.ones_like(z))
    ref = torch.logsumexp(w, 0) - math.log(K)
    m_run, s_run, ll = (torch.empty(N, device=dev) for _ in range(3))
    d = [t.to(dev) for t in (rl, eps, mu, sg)]
    nv.call("btvae_iw_lse_update", P(d[0][:4].contiguous()), P(d[1][:4].contiguous()), P(d[2]), P(d[3]), P(m_run), P(s_run), 4, N, L, 1)
    nv.call("btvae_iw_lse_update", P(d[0][4:].contiguous()), P(d[1][4:].contiguous()), P(d[2]), P(d[3]), P(m_run), P(s_run), K - 4, N, L, 0)
    nv.call("btvae_iw_lse_finalize", P(m_run), P(s_run), P(ll), N, K)
    close(ll, ref, 1e-5, 1e-4, "iw lse")


def test_clip_adam(dev):
    n = 100003
    p0, g = rnd(n, seed=1).float(), (rnd(n, seed=2) * 0.05).float()
    pr = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=1e-5)
    # arenas padded to a multiple of 4 floats, 16-byte aligned
    p, gd, m, v = (torch.zeros(n + 1, device=dev) for _ in range(4))
    p[:n] = p0.to(dev)
    part = torch.empty(512, dtype=torch.float64, device=dev)
    nrm = torch.empty(1, device=dev)
    for step in range(1, 4):
        gs = g * step
        pr.grad = gs.double().clone()
        tot = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd[:n] = gs.to(dev)
        nv.call("btvae_grad_sumsq", P(gd), n, P(part))
        nv.call("btvae_clip_adam_step", P(p), P(gd), P(m), P(v), n, P(part), 1.0, 1e-3, 0.9, 0.999, 1e-8, 1e-5, step, P(nrm))
        close(nrm, tot.reshape(1), 1e-6, 0, "norm")
        close(p[:n], pr.detach(), 1e-6, 2e-7, f"params step {step}")


@pytest.mark.parametrize("rows,D,obs", [(70, 48, list(range(0, 12)) + list(range(36, 48))),     # a non-suffix window
                                        (5, 64, list(range(1, 64, 2))),                          # largest dimension, interleaved
                                        (9, 7, [6]),                                             # one observed coordinate
                                        (3, 24, list(range(23))),                                # one unobserved coordinate
                                        (11, 96, list(range(0, 96, 2))),                         # beyond 64: the LDS kernel, two mask words
                                        (4, 128, list(range(60, 128))),                          # observed block across the word boundary
                                        (5, 65, [64]),                                           # one observed coordinate, in the second word
                                        (6, 40, list(range(0, 40, 3)))])                         # the two-word entry point forwarding to the register kernel
def test_cond_mvn(dev, rows, D, obs):
    Uc = D - len(obs)
    NP = D * (D + 1) // 2
    A = rnd(rows, D, D, seed=1) * 0.3
    Sigma = A @ A.transpose(-1, -2) + torch.eye(D, dtype=torch.float64) * 0.01
    Sigma = Sigma.float().double()
    mu, x, eps = rnd(rows, D, seed=2).float(), rnd(rows, D, seed=3).float(), rnd(rows, Uc, seed=4).float()
    un = [i for i in range(D) if i not in obs]
    tri = torch.tril_indices(D, D)
    Sp = Sigma[:, tri[0], tri[1]].float().to(dev)
    mask = sum(1 << i for i in obs)
    un_pos = torch.zeros(D, dtype=torch.int32)
    for r_, i in enumerate(un):
        un_pos[i] = r_
    samp, cm, cc = torch.empty(rows, Uc, device=dev), torch.empty(rows, Uc, device=dev), torch.empty(rows, Uc, Uc, device=dev)
    mud, xd, epsd, upd = mu.to(dev), x.to(dev), eps.to(dev), un_pos.to(dev)
    if D > 64 or D == 40:
        import ctypes
        words = (ctypes.c_ulonglong * 2)(mask & (2 ** 64 - 1), mask >> 64)
        nv.call("btvae_cond_mvn_sample_wide", P(Sp), P(mud), P(xd), P(epsd), ctypes.cast(words, ctypes.c_void_p), P(upd), P(samp), P(cm), P(cc),
                rows, D, Uc)
    else:
        nv.call("btvae_cond_mvn_sample", P(Sp), P(mud), P(xd), P(epsd), mask, P(upd), P(samp), P(cm), P(cc), rows, D, Uc)
    S64 = Sp.cpu().double()
    full = torch.zeros(rows, D, D, dtype=torch.float64)
    full[:, tri[0], tri[1]] = S64
    full = full + full.transpose(-1, -2) - torch.diag_embed(torch.diagonal(full, dim1=-2, dim2=-1))
    Soo, Suo, Suu = full[:, obs][:, :, obs], full[:, un][:, :, obs], full[:, un][:, :, un]
    Aco = Suo @ torch.linalg.inv(Soo)
    mean = mu.double()[:, un] + (Aco @ (x.double()[:, obs] - mu.double()[:, obs])[..., None])[..., 0]
    cov = Suu - Aco @ Suo.transpose(-1, -2)
    close(cm, mean, 2e-4, 2e-4, "cond mean")
    close(cc, cov, 2e-4, 2e-5, "cond cov")
    sref = mean + (torch.linalg.cholesky(cov) @ eps.double()[..., None])[..., 0]
    close(samp, sref, 2e-4, 2e-4, "cond sample")
    # scatter + pack + lse
    dst = torch.zeros(rows, D, device=dev)
    idx = (torch.arange(rows)[:, None] * D + torch.tensor(un)[None, :]).reshape(-1).to(dev)
    nv.call("btvae_masked_scatter", P(dst), P(idx), P(samp), rows * Uc)
    assert torch.equal(dst[:, un], samp) and torch.count_nonzero(dst[:, obs]) == 0
    pk = torch.empty(rows, Uc * (Uc + 1) // 2, device=dev)
    nv.call("btvae_pack_lower", P(cc), P(pk), rows, Uc)
    t2 = torch.tril_indices(Uc, Uc)
    assert torch.equal(pk, cc[:, t2[0], t2[1]])
    lp = (rnd(7, rows, seed=9) * 10).float().to(dev)
    o = torch.empty(rows, device=dev)
    nv.call("btvae_lse_over_k", P(lp), P(o), 7, rows)
    close(o, torch.logsumexp(lp.cpu().double(), 0) - math.log(7), 1e-6, 1e-5, "lse over k")


@pytest.mark.parametrize("M,N,K", [(300, 200, 136), (128, 128, 32), (1000, 48, 1000), (257, 100, 52)])
def test_linear_products_bf16_operands(dev, M, N, K):
    """bf16-operand mode of the Linear products: equals the fp64 product of the bf16-rounded operands (fp32
    accumulation error only), for forward (+bias, ReLU, column statistics), dgrad (+add-in) and split wgrad."""
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * 0.1; b = torch.randn(N, generator=g)
    dY = torch.randn(M, N, generator=g); R = torch.randn(M, K, generator=g)
    rb = lambda t: t.to(torch.bfloat16).double()
    Xd, Wd, bd, dYd, Rd = (t.to(dev) for t in (X, W, b, dY, R))
    tiles = (M + 127) // 128
    Y = torch.empty(M, N, device=dev); part = torch.empty(tiles, 2, N, device=dev)
    dX = torch.empty(M, K, device=dev); dW = torch.empty(N, K, device=dev)
    ws = torch.empty(4 * 1024 * 1024, device=dev)
    nv.set_gemm_precision("bf16")
    try:
        assert nv.get_gemm_precision() == "bf16"
        nv.call("btvae_linear_relu_stats_fwd", P(Xd), K, P(Wd), P(bd), P(Y), P(part), M, K, N)
        nv.call("btvae_linear_dgrad", P(dYd), P(Wd), P(Rd), P(dX), K, M, N, K)
        nv.call("btvae_linear_wgrad", P(dYd), P(Xd), K, P(dW), P(ws), ws.numel(), M, N, K, 0)
        torch.cuda.synchronize()
    finally:
        nv.set_gemm_precision("fp32")
    ref_y = torch.relu(rb(X) @ rb(W).t() + b.double())
    assert (Y.cpu().double() - ref_y).abs().max() <= 2e-5 * max(1.0, ref_y.abs().max().item())
    ref_dx = rb(dY) @ rb(W) + R.double()
    assert (dX.cpu().double() - ref_dx).abs().max() <= 2e-5 * ref_dx.abs().max().item()
    ref_dw = rb(dY).t() @ rb(X)
    assert (dW.cpu().double() - ref_dw).abs().max() <= 2e-5 * ref_dw.abs().max().item()
    # column statistics partials describe the Y that was written
    yy = Y.cpu().double()
    for t in range(tiles):
        blk = yy[t * 128:(t + 1) * 128]
        assert torch.allclose(part[t, 0].cpu().double(), blk.mean(0), rtol=1e-4, atol=1e-5)
        assert torch.allclose(part[t, 1].cpu().double(), ((blk - blk.mean(0)) ** 2).sum(0), rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("M,H", [(300, 200), (520, 72), (16500, 1000)])
def test_bf16_stored_hidden_matrices(dev, M, H):
    """bf16 mode with the hidden matrices stored as bf16 (``btvae_next_bf16``): every kernel that reads or writes such a
    matrix against fp64 on the bf16 values it was given -- the forward, eval-mode forward, dX (plain, with the BatchNorm
    backward sums, with the fused BatchNorm backward) and weight-gradient products, with the weight's stage images (the
    256-wide DMA-fed kernel: 128-row workgroups, and 256-row ones at M = 16,500) and without (the 128-wide kernel); the
    head forward; the BatchNorm apply / backward kernels.  Outputs are bf16: compared to half a unit in the last place of
    bf16 (2^-8 relative) plus the fp32 accumulation error; pad columns of every output row are written as zeros."""
    L = nv.lib()
    g = torch.Generator().manual_seed(M + H)
    ld = (H + 31) // 32 * 32
    K0, V = 52, 100
    A, OUT, AUX, B = 1, 2, 4, 8

    def hid(t):                                   # fp32 [M, H] -> bf16 matrix of pitch ld, pad columns zero
        buf = torch.zeros(t.shape[0], ld, dtype=torch.bfloat16, device=dev)
        buf[:, :H] = t.to(dev)
        return buf

    def out16():                                  # an output buffer whose pad columns must come back as zeros
        return torch.full((M, ld), float("nan"), dtype=torch.bfloat16, device=dev)

    def check(got16, ref, what, extra=0.0):
        got = got16[:, :H].double().cpu()
        lim = 2.0 ** -8 * ref.abs() + 1e-5 * ref.abs().max() + extra
        assert ((got - ref).abs() <= lim).all(), (what, float(((got - ref).abs() - lim).max()))
        if ld > H:
            assert bool((got16[:, H:] == 0).all()), what + ": pad columns"

    X = torch.randn(M, H, generator=g).clamp_(min=0); W = torch.randn(H, H, generator=g) * (1.0 / H ** 0.5); b = torch.randn(H, generator=g) * 0.1
    Xb = hid(X); Xr = Xb[:, :H].double().cpu(); Wr = W.to(torch.bfloat16).double()
    Wd, bd = W.to(dev), b.to(dev)
    tiles = (M + 127) // 128
    part, part2 = torch.empty(tiles, 2, H, device=dev), torch.empty(tiles, 2, H, device=dev)
    img = torch.empty(L.btvae_weight_image_bf16_bytes(H, H, 0) + L.btvae_weight_image_bf16_bytes(H, H, 1), dtype=torch.uint8, device=dev)
    nv.set_gemm_precision("bf16")
    try:
        nv.call("btvae_weight_image_bf16", P(Wd), H, H, H, 3, P(img))
        # ---- forward: relu(X W^T + b) with column statistics
        ref = torch.relu(Xr @ Wr.t() + b.double())
        for planes, pt in ((img, part), (None, part2)):
            Y = out16()
            L.btvae_next_row_pitches(ld, 0); L.btvae_next_bf16(A | OUT)
            nv.call("btvae_linear_relu_stats_fwd_pw", P(Xb), ld, P(Wd), P(planes), P(bd), P(Y), P(pt), M, H, H)
            check(Y, ref, "forward" + (" (images)" if planes is not None else ""))
        assert torch.allclose(part, part2, rtol=1e-4, atol=1e-5)          # the statistics are taken before the rounding
        t0 = ref[:128]
        assert torch.allclose(part[0, 0].double().cpu(), t0.mean(0), rtol=1e-3, atol=1e-4)
        # ---- a first layer: fp32 input, bf16 output
        X0 = torch.randn(M, K0, generator=g); W0 = torch.randn(H, K0, generator=g) * 0.2
        X0d, W0d = X0.to(dev), W0.to(dev)
        Y = out16()
        L.btvae_next_row_pitches(ld, 0); L.btvae_next_bf16(OUT)
        nv.call("btvae_linear_relu_stats_fwd", P(X0d), K0, P(W0d), P(bd), P(Y), P(part2), M, K0, H)
        check(Y, torch.relu(X0.to(torch.bfloat16).double() @ W0.to(torch.bfloat16).double().t() + b.double()), "first layer")
        # ---- eval mode: relu(.) * scale + shift + X
        sc, sh = (1 + 0.2 * torch.randn(H, generator=g)), 0.1 * torch.randn(H, generator=g)
        scd, shd = sc.to(dev), sh.to(dev)
        ref = torch.relu(Xr @ Wr.t() + b.double()) * sc.double() + sh.double() + Xr
        for planes in (img, None):
            Y = out16()
            L.btvae_next_bf16(A | OUT | AUX)
            nv.call("btvae_linear_relu_affine_fwd_pw_ld", P(Xb), ld, P(Wd), P(planes), P(bd), P(scd), P(shd), P(Xb), ld, P(Y), ld, M, H, H)
            check(Y, ref, "eval forward")
        # ---- dX products: G = dY W (+ R)
        dY = torch.randn(M, H, generator=g); R = torch.randn(M, H, generator=g)
        dYb, Rb = hid(dY), hid(R)
        dYr, Rr = dYb[:, :H].double().cpu(), Rb[:, :H].double().cpu()
        Gref = dYr @ Wr + Rr
        for planes in (img, None):
            G = out16()
            L.btvae_next_row_pitches(ld, 0); L.btvae_next_bf16(A | OUT | AUX)
            nv.call("btvae_linear_dgrad_pw", P(dYb), P(Wd), P(planes), P(Rb), P(G), ld, M, H, H)
            check(G, Gref, "dgrad")
        mean, inv = Xr.mean(0), 1 / torch.sqrt(Xr.var(0, unbiased=False) + 1e-5)
        gam = 1 + 0.3 * torch.randn(H, generator=g)
        md, idv, gd = mean.float().to(dev), inv.float().to(dev), gam.to(dev)
        xhat = (Xr - mean) * inv
        for planes in (img, None):
            G = out16()
            gp = torch.empty(tiles, 2, H, device=dev)
            L.btvae_next_row_pitches(ld, ld); L.btvae_next_bf16(A | OUT | AUX)
            nv.call("btvae_linear_dgrad_gstats_pw", P(dYb), P(Wd), P(planes), P(Rb), P(Xb), P(md), P(idv), P(G), P(gp), M, H, H)
            check(G, Gref, "dgrad + sums")
            sums = gp.double().cpu().sum(0)
            assert torch.allclose(sums[0], Gref.sum(0), rtol=2e-3, atol=2e-3 * float(Gref.abs().sum(0).max()))
            assert torch.allclose(sums[1], (Gref * xhat).sum(0), rtol=2e-3, atol=2e-3 * float((Gref * xhat).abs().sum(0).max()))
            s2 = torch.stack([Gref.sum(0), (Gref * xhat).sum(0)]).float().to(dev)
            dP = out16()
            cp = torch.empty(tiles, H, device=dev)
            L.btvae_next_row_pitches(ld, ld); L.btvae_next_bf16(A | OUT | AUX)
            nv.call("btvae_linear_dgrad_bnbwd_pw", P(dYb), P(Wd), P(planes), P(Rb), P(Xb), P(md), P(idv), P(gd), P(s2), M, P(dP), P(cp), M, H, H)
            gi = (gam.double() * inv)
            dref = torch.where(Xr > 0, gi * (Gref - s2[0].double().cpu() / M - xhat * s2[1].double().cpu() / M), torch.zeros_like(Gref))
            check(dP, dref, "dgrad + BatchNorm backward", extra=2.0 ** -8 * float(gi.max()) * Gref.abs())
        # ---- weight gradients: dW = dY^T X in the three storage combinations
        ws = torch.empty(8 * 1024 * 1024, device=dev)
        dW = torch.empty(H, H, device=dev)
        for a16, b16 in ((True, True), (False, True), (True, False)):
            da = dYb if a16 else dYb[:, :H].float().contiguous()
            xb = Xb if b16 else Xb[:, :H].float().contiguous()
            L.btvae_next_row_pitches(da.stride(0) if a16 else 0, 0); L.btvae_next_bf16((A if a16 else 0) | (B if b16 else 0))
            nv.call("btvae_linear_wgrad", P(da), P(xb), xb.stride(0), P(dW), P(ws), ws.numel(), M, H, H, 0)
            ref = dYr.t() @ Xr
            assert float((dW.double().cpu() - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-4, (a16, b16)
        # ---- head output layer on a bf16 hidden activation
        Wh, bh = torch.randn(V, H, generator=g) * 0.05, torch.randn(V, generator=g) * 0.1
        Whd, bhd = Wh.to(dev), bh.to(dev)
        val, dfac = torch.empty(M, V, device=dev), torch.empty(M, V, device=dev)
        L.btvae_next_bf16(A)
        nv.call("btvae_linear_head_fwd", P(Xb), ld, P(Whd), P(bhd), 2, 1e-4, 0.62, P(val), P(dfac), None, M, H, V)
        ref = torch.relu(Xr @ Wh.to(torch.bfloat16).double().t() + bh.double()).clamp(1e-4, 0.62)
        assert float((val.double().cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    finally:
        nv.set_gemm_precision("fp32")
    # ---- BatchNorm kernels on bf16 matrices (the storage flag is independent of the GEMM mode)
    beta = 0.1 * torch.randn(H, generator=g)
    betad = beta.to(dev)
    Out = out16()
    L.btvae_next_row_pitches(ld, 0); L.btvae_next_bf16(OUT)
    nv.call("btvae_bn_apply_fwd", P(Xb), P(md), P(idv), P(gd), P(betad), P(Rb), None, 1.0, P(Out), M, H)
    check(Out, xhat * gam.double() + beta.double() + Rr, "bn_apply_fwd")
    scr = torch.empty(L.btvae_bn_relu_bwd_scratch(M, H), device=dev)
    dg, dbt, dbi = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    dP = out16()
    L.btvae_next_row_pitches(ld, 0); L.btvae_next_bf16(OUT)
    nv.call("btvae_bn_relu_bwd", P(dYb), P(Xb), P(md), P(idv), P(gd), None, 1.0, P(dg), P(dbt), P(dbi), None, P(dP), P(scr), M, H, 0)
    s1, s2 = dYr.sum(0), (dYr * xhat).sum(0)
    dref = torch.where(Xr > 0, (gam.double() * inv) * (dYr - s1 / M - xhat * s2 / M), torch.zeros_like(dYr))
    check(dP, dref, "bn_relu_bwd", extra=1e-3 * float(dref.abs().max()))
    assert torch.allclose(dbt.double().cpu(), s1, rtol=1e-3, atol=1e-3 * float(dYr.abs().sum(0).max()))
    assert torch.allclose(dg.double().cpu(), s2, rtol=1e-3, atol=1e-3 * float((dYr * xhat).abs().sum(0).max()))
    dP = out16()
    L.btvae_next_row_pitches(ld, 0); L.btvae_next_bf16(OUT)
    nv.call("btvae_relu_bwd", P(dYb), P(Xb), None, 1.0, P(dbi), None, P(dP), P(scr), M, H, 0)
    check(dP, torch.where(Xr > 0, dYr, torch.zeros_like(dYr)), "relu_bwd")


def test_linear_products_at_benchmark_size(dev, gemm_mode):
    """The three Linear products at the benchmark's decoder shape (M = S*B = 65536, N = K = 1000): forward rows and
    dgrad rows sampled against fp64 on the CPU (2e-6 relative + 2e-6*sqrt(reduction length) absolute, as at small
    sizes), BatchNorm statistics and the whole weight gradient against fp64 products on the device."""
    M, N, K = 65536, 1000, 1000
    g = torch.Generator(device=dev).manual_seed(11)
    X = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.03
    b = torch.randn(N, device=dev, generator=g)
    Y = torch.empty(M, N, device=dev)
    part = torch.empty((M + 127) // 128, 2, N, device=dev)
    nv.call("btvae_linear_relu_stats_fwd", P(X), K, P(W), P(b), P(Y), P(part), M, K, N)
    rows = torch.randint(0, M, (96,), generator=torch.Generator().manual_seed(3)).tolist() + [0, 127, 128, M - 1]
    ref = torch.relu(X[rows].cpu().double() @ W.cpu().double().t() + b.cpu().double())
    close(Y[rows], ref, 2e-6, 2e-6 * math.sqrt(K), "forward rows")
    mean, inv = torch.empty(N, device=dev), torch.empty(N, device=dev)
    nv.call("btvae_bn_stats_finalize", P(part), (M + 127) // 128, M, N, P(mean), P(inv), None, None, 0.1, 1e-5)
    Yd = Y.double()
    close(mean, Yd.mean(0), 1e-5, 1e-6, "column mean")
    close(inv, 1.0 / torch.sqrt(Yd.var(0, unbiased=False) + 1e-5), 1e-5, 0.0, "column invstd")
    dY = torch.randn(M, N, device=dev, generator=g)
    dX = torch.empty(M, K, device=dev)
    nv.call("btvae_linear_dgrad", P(dY), P(W), None, P(dX), K, M, N, K)
    close(dX[rows], dY[rows].cpu().double() @ W.cpu().double(), 2e-6, 2e-6 * math.sqrt(N), "dgrad rows")
    dW = torch.empty(N, K, device=dev)
    ws = torch.empty(32 * 1024 * 1024, device=dev)
    nv.call("btvae_linear_wgrad", P(dY), P(X), K, P(dW), P(ws), ws.numel(), M, N, K, 0)
    # sums of 65536 unit-variance products (result rms 256) accumulated in fp32, 4096 terms per split: every add rounds
    # a running sum of magnitude ~64, a random walk of a few 1e-4 per split; 1.2e-5 of the result's rms is the bound
    close(dW, dY.double().t() @ X.double(), 2e-6, 1.2e-5 * math.sqrt(M), "wgrad")


@pytest.mark.parametrize("M,K0,H", [(300, 28, 200), (520, 52, 72), (16500, 28, 1000)])
def test_pair_format_eval_chain(dev, M, K0, H):
    """Eval-mode units with their activations in the pair format (``btvae_linear_relu_affine_fwd_pairs``, csrc/pairs.h):
    fp32 input -> pairs, pairs -> pairs (plain and residual), pairs -> fp32, each against fp64 on the values it was given; 128-row workgroups
    and (M = 16,500) 256-row ones with a row group wholly past the matrix; pad columns of a pair-format row come back as
    zeros; whole-matrix encode / decode.  Tolerance: the two-piece product's (2e-6 relative + 2e-6 sqrt(K) of the operand
    scale) plus, for a pair-format output, 2^-22 of its block's maximum."""
    nv.set_gemm_precision("fp32_split2")
    try:
        L = nv.lib()
        g = torch.Generator().manual_seed(M + H)
        ld = (H + 31) // 32 * 32
        X0 = torch.randn(M, K0, generator=g).to(dev)
        Ws = [(torch.randn(H, K0, generator=g) / math.sqrt(K0)).to(dev), (torch.randn(H, H, generator=g) / math.sqrt(H)).to(dev),
              (torch.randn(H, H, generator=g) / math.sqrt(H)).to(dev)]
        bs = [(0.3 * torch.randn(H, generator=g)).to(dev) for _ in range(3)]
        scs = [(0.5 + torch.rand(H, generator=g)).to(dev) for _ in range(3)]
        shs = [(0.2 * torch.randn(H, generator=g)).to(dev) for _ in range(3)]

        def slot(t):
            s = torch.zeros(2, device=dev)
            nv.call("btvae_absmax", P(t), t.numel(), P(s))
            return s

        def ref(x64, i):
            return torch.relu(x64 @ Ws[i].double().t() + bs[i].double()) * scs[i].double() + shs[i].double()

        def pair_buffer():
            return torch.full((M, ld), float("nan"), device=dev), torch.full(((M + 127) // 128, (H + 127) // 128), float("nan"), device=dev)

        def decode(buf, sc, width):
            out = torch.empty(M, ld, device=dev)
            nv.call("btvae_pairs_decode", P(buf), ld, P(sc), P(out), ld, M, width)
            return out

        # unit 1: fp32 in, pairs out (128-wide two-piece kernel, weight as fp16 planes)
        xmax, wmax = slot(X0), slot(Ws[0])
        planes = torch.empty(L.btvae_weight_planes_elems(H, K0), dtype=torch.int16, device=dev)
        nv.call("btvae_weight_planes_f16", P(Ws[0]), K0, H, K0, P(planes), 1, P(wmax))
        y1, s1 = pair_buffer()
        L.btvae_gemm_operand_maxima(P(xmax), P(wmax))
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(X0), K0, None, P(Ws[0]), P(planes), None, P(bs[0]), P(scs[0]), P(shs[0]), 0,
                P(y1), ld, P(s1), M, K0, H)
        torch.cuda.synchronize()
        assert torch.isfinite(s1).all() and (s1 > 0).all()
        full = decode(y1, s1, ld)                       # every column up to the pitch: the pad columns must decode to zero
        assert torch.equal(full[:, H:], torch.zeros_like(full[:, H:])), "pad columns of a pair-format row"
        r1 = ref(X0.double(), 0)
        blockmax = float(r1.abs().max())
        close(full[:, :H], r1, 2e-6, 2e-6 * math.sqrt(K0) * float(X0.abs().max()) + 2.0 ** -22 * blockmax, "fp32 -> pairs")

        # unit 2: pairs in, pairs out (256-wide DMA-fed kernel)
        x2 = full[:, :H].double()                       # the values the pair-format operand holds
        wmax2 = slot(Ws[1])
        img = torch.empty(L.btvae_weight_image_pq_bytes(H, H, 0), dtype=torch.uint8, device=dev)
        nv.call("btvae_weight_image_pq", P(Ws[1]), H, H, H, 0, P(img), P(wmax2))
        y2, s2 = pair_buffer()
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(y1), ld, P(s1), None, P(img), P(wmax2), P(bs[1]), P(scs[1]), P(shs[1]), 0,
                P(y2), ld, P(s2), M, H, H)
        full2 = decode(y2, s2, ld)
        assert torch.equal(full2[:, H:], torch.zeros_like(full2[:, H:]))
        r2 = ref(x2, 1)
        close(full2[:, :H], r2, 2e-6, 2e-6 * math.sqrt(H) * float(x2.abs().max()) + 2.0 ** -22 * float(r2.abs().max()), "pairs -> pairs")

        # the residual form: the pair-format input is added to the result (its block scale comes from a bound on the sum)
        y2r, s2r = pair_buffer()
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(y1), ld, P(s1), None, P(img), P(wmax2), P(bs[1]), P(scs[1]), P(shs[1]), 1,
                P(y2r), ld, P(s2r), M, H, H)
        full2r = decode(y2r, s2r, ld)
        assert torch.equal(full2r[:, H:], torch.zeros_like(full2r[:, H:]))
        r2r = r2 + x2
        close(full2r[:, :H], r2r, 2e-6, 2e-6 * math.sqrt(H) * float(x2.abs().max()) + 2.0 ** -20 * float(r2r.abs().max()), "pairs -> pairs, residual")

        # unit 3: pairs in, fp32 out
        wmax3 = slot(Ws[2])
        img3 = torch.empty(L.btvae_weight_image_pq_bytes(H, H, 0), dtype=torch.uint8, device=dev)
        nv.call("btvae_weight_image_pq", P(Ws[2]), H, H, H, 0, P(img3), P(wmax3))
        y3 = torch.full((M, ld), float("nan"), device=dev)
        nv.call("btvae_linear_relu_affine_fwd_pairs", P(y2), ld, P(s2), None, P(img3), P(wmax3), P(bs[2]), P(scs[2]), P(shs[2]), 0,
                P(y3), ld, None, M, H, H)
        x3 = full2[:, :H].double()
        close(y3[:, :H], ref(x3, 2), 2e-6, 2e-6 * math.sqrt(H) * float(x3.abs().max()), "pairs -> fp32")

        # whole-matrix conversion: encode, decode, compare (22 bits below each block's maximum)
        Xe = torch.zeros(M, ld, device=dev)
        Xe[:, :H] = torch.randn(M, H, generator=g).to(dev) * torch.logspace(-3, 3, H).to(dev)      # block maxima spread over six decades
        Pe, se = pair_buffer()
        Hc = (H + 7) // 8 * 8
        nv.call("btvae_pairs_encode", P(Xe), ld, P(Pe), ld, P(se), M, Hc)
        back = decode(Pe, se, Hc)
        for bi in range((M + 127) // 128):
            for bj in range((H + 127) // 128):
                blk = Xe[bi * 128:(bi + 1) * 128, bj * 128:min((bj + 1) * 128, H)]
                got = back[bi * 128:(bi + 1) * 128, bj * 128:min((bj + 1) * 128, H)]
                assert float((got - blk).abs().max()) <= 2.0 ** -22 * float(blk.abs().max()) + 1e-30, (bi, bj)
    finally:
        nv.set_gemm_precision("fp32")


@pytest.mark.parametrize("M,H", [(300, 200), (520, 72), (16500, 1000)])
def test_pair_format_training_side(dev, M, H):
    """The train-mode entry points on pair-format activations: the BatchNorm-apply pass that writes one (no add-in, an fp32
    add-in, a pair-format add-in), the Linear+ReLU+statistics product that reads one, the weight gradient that reads one.
    Each against fp64 on the values the pair-format operand holds (its decode).  Tolerances: a pair-format output 2^-22 of its
    block's maximum (2^-20 with an add-in: the scale comes from a bound on the sum); the products as their fp32-operand
    forms."""
    nv.set_gemm_precision("fp32_split2")
    try:
        L = nv.lib()
        g = torch.Generator().manual_seed(7 * M + H)
        ld = (H + 31) // 32 * 32
        nb = ((M + 127) // 128, (H + 127) // 128)

        def padded(t):
            buf = torch.zeros(M, ld, device=dev)
            buf[:, :H] = t.to(dev)
            return buf

        def pair_buffer():
            return torch.full((M, ld), float("nan"), device=dev), torch.full(nb, float("nan"), device=dev)

        def decode(buf, sc):
            out = torch.empty(M, ld, device=dev)
            nv.call("btvae_pairs_decode", P(buf), ld, P(sc), P(out), ld, M, ld)
            return out

        def slot(t):
            s = torch.zeros(2, device=dev)
            nv.call("btvae_absmax", P(t), t.numel(), P(s))
            return s

        # BatchNorm-apply writing pairs
        Y = padded(torch.relu(torch.randn(M, H, generator=g)) * torch.logspace(-1, 1, H))
        mean, invstd = (0.3 * torch.randn(H, generator=g)).to(dev), (0.5 + torch.rand(H, generator=g)).to(dev)
        gamma, beta = (0.5 + torch.rand(H, generator=g)).to(dev), (0.2 * torch.randn(H, generator=g)).to(dev)
        sc64 = (invstd * gamma).double()
        ref0 = (Y[:, :H].double() - mean.double()) * sc64 + beta.double()
        o0, s0 = pair_buffer()
        nv.call("btvae_bn_apply_fwd_pairs", P(Y), ld, P(mean), P(invstd), P(gamma), P(beta), None, 0, None, P(o0), ld, P(s0), M, H)
        d0 = decode(o0, s0)
        assert torch.equal(d0[:, H:], torch.zeros_like(d0[:, H:])), "pad columns of a pair-format row"
        close(d0[:, :H], ref0, 4e-7, 2.0 ** -22 * float(ref0.abs().max()), "bn apply -> pairs")
        R = padded(torch.randn(M, H, generator=g))
        o1, s1 = pair_buffer()
        nv.call("btvae_bn_apply_fwd_pairs", P(Y), ld, P(mean), P(invstd), P(gamma), P(beta), P(R), ld, None, P(o1), ld, P(s1), M, H)
        d1 = decode(o1, s1)
        ref1 = ref0 + R[:, :H].double()
        close(d1[:, :H], ref1, 4e-7, 2.0 ** -20 * float(ref1.abs().max()), "bn apply + fp32 add-in -> pairs")
        o2, s2 = pair_buffer()
        nv.call("btvae_bn_apply_fwd_pairs", P(Y), ld, P(mean), P(invstd), P(gamma), P(beta), P(o0), ld, P(s0), P(o2), ld, P(s2), M, H)
        d2 = decode(o2, s2)
        ref2 = ref0 + d0[:, :H].double()
        assert torch.equal(d2[:, H:], torch.zeros_like(d2[:, H:]))
        close(d2[:, :H], ref2, 4e-7, 2.0 ** -20 * float(ref2.abs().max()), "bn apply + pair-format add-in -> pairs")
        # the pass without BatchNorm vectors is a plain conversion (+ add-in)
        o3, s3 = pair_buffer()
        nv.call("btvae_bn_apply_fwd_pairs", P(Y), ld, None, None, None, None, None, 0, None, P(o3), ld, P(s3), M, H)
        close(decode(o3, s3)[:, :H], Y[:, :H], 0.0, 2.0 ** -22 * float(Y.abs().max()), "conversion")

        # Linear + ReLU + statistics on the pair-format matrix o1
        xv = d1[:, :H].double().cpu()
        N = H
        W = (torch.randn(N, H, generator=g) / math.sqrt(H)).to(dev)
        b = (0.3 * torch.randn(N, generator=g)).to(dev)
        wmax = slot(W)
        img = torch.empty(L.btvae_weight_image_pq_bytes(N, H, 0), dtype=torch.uint8, device=dev)
        nv.call("btvae_weight_image_pq", P(W), H, N, H, 0, P(img), P(wmax))
        tiles = (M + 127) // 128
        Yo = torch.full((M, ld), float("nan"), device=dev)
        part = torch.empty(tiles, 2, N, device=dev)
        nv.call("btvae_linear_relu_stats_fwd_pairs", P(o1), ld, P(s1), P(img), P(wmax), P(b), P(Yo), ld, P(part), M, H, N)
        yref = torch.relu(xv @ W.cpu().double().t() + b.cpu().double())
        close(Yo[:, :N], yref, 2e-6, 2e-6 * math.sqrt(H) * float(xv.abs().max()), "pairs -> relu(linear)")
        mu, isd = torch.empty(N, device=dev), torch.empty(N, device=dev)
        nv.call("btvae_bn_stats_finalize", P(part), tiles, M, N, P(mu), P(isd), None, None, 0.1, 1e-5)
        yr = Yo[:, :N].cpu().double()
        close(mu, yr.mean(0), 1e-5, 1e-6 * float(yr.abs().max()), "mean")
        close(isd, 1 / torch.sqrt(yr.var(0, unbiased=False) + 1e-5), 2e-5, 0, "invstd")

        # weight gradient with the pair-format matrix as the B operand
        Nw = 72 if H != 72 else 200
        ldn = (Nw + 31) // 32 * 32
        dY = torch.zeros(M, ldn, device=dev)
        dY[:, :Nw] = (torch.randn(M, Nw, generator=g) * torch.logspace(-2, 0, Nw)).to(dev)
        dmax = slot(dY)
        dW = torch.full((Nw, H), float("nan"), device=dev)
        ws = torch.empty(8 * 1024 * 1024, device=dev)
        for acc in (0, 1):
            L.btvae_gemm_operand_maxima(P(dmax), None)
            L.btvae_next_row_pitches(ldn, 0)
            nv.call("btvae_linear_wgrad_pairs", P(dY), P(o1), ld, P(s1), P(dW), P(ws), ws.numel(), M, Nw, H, acc)
        wref = 2.0 * (dY[:, :Nw].cpu().double().t() @ xv)
        close(dW, wref, 2e-6, 4e-6 * math.sqrt(M) * float(dY.abs().max()) * float(xv.abs().max()) / 4.0, "wgrad on pairs")
    finally:
        nv.set_gemm_precision("fp32")
