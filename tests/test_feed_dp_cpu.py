"""CPU: condition feed against the reference's golden outputs, the device batch loader's index stream against
torch's own RandomSampler, and the data-parallel plumbing with world_size 2 on gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def make_conditioner(g, style):
    from behavio_temporal_vae_amd.conditioning_lib import Conditioner
    cset = {k: g[k] for k in ("months", "weekdays", "day_befores", "users")}
    cond = Conditioner(tags=["months", "weekdays", "day_befores", "users"], supports=[list(range(1, 13)), list(range(7)), [-3, 3], [0.01, 10.0]],
                       types=["circ", "circ", "identity", "dir"], condition_set=cset)
    cond.transformers["users"].transform_style = style
    return cond, cset


@pytest.mark.parametrize("style", ["mean", "embed", "scaled", "sample"])
def test_conditioner_matches_reference(style):
    g = load_golden("conditioner")
    cond, cset = make_conditioner(g, style)
    np.random.seed(4)
    out = cond.transform(cset, num_samples=3)
    assert out.shape == g["out_" + style].shape
    assert np.allclose(out, g["out_" + style], rtol=1e-12, atol=1e-12)
    assert cond.is_random() == (style == "sample")


def test_condition_feed_layouts():
    from behavio_temporal_vae_amd.datasets import ConditionFeed, ConditionedDataset
    g = load_golden("conditioner")
    x = np.arange(12 * 3, dtype=np.float32).reshape(12, 3)
    cond, cset = make_conditioner(g, "mean")
    feed = list(ConditionFeed(x, cset, cond, batch_size=5))
    assert [(b0, b1) for b0, b1, _, _ in feed] == [(0, 5), (5, 10), (10, 12)]
    full = cond.transform(cset).astype(np.float32)
    assert all(np.array_equal(c.numpy(), full[b0:b1]) for b0, b1, _, c in feed)
    # per-item dataset interface agrees with the vectorised feed
    ds = ConditionedDataset(x, cset, cond)
    xi, ci = ds[7]
    assert np.array_equal(xi.numpy(), x[7]) and np.allclose(ci.numpy(), full[7])
    cond_s, _ = make_conditioner(g, "sample")
    fs = list(ConditionFeed(x, cset, cond_s, batch_size=5, num_samples=4))
    assert fs[0][3].shape == (5, 4, cond_s.cond_dim) and fs[2][3].shape == (2, 4, cond_s.cond_dim)


def test_device_loader_index_stream_matches_torch_sampler():
    from behavio_temporal_vae_amd.loaders import DeviceBatchLoader, shard_range
    from behavio_temporal_vae_amd.synthetic import calendar_conditioner, calendar_conditions
    from behavio_temporal_vae_amd.datasets import ConditionedDataset
    rng = np.random.default_rng(0)
    n, B = 103, 10
    cset = calendar_conditions(n, rng)
    ds = ConditionedDataset(np.arange(n, dtype=np.float32)[:, None].repeat(2, 1), cset, calendar_conditioner(cset))
    torch.manual_seed(123)
    ref = [b[0][:, 0].long().tolist() for b in torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True, drop_last=True)]
    torch.manual_seed(123)
    got = [x[:, 0].long().tolist() for x, _ in DeviceBatchLoader(ds, B, shuffle=True, drop_last=True, device="cpu")]
    assert got == ref and len(got) == n // B
    # two ranks split every global batch of 2B contiguously and together cover it
    torch.manual_seed(123)
    glob = [x[:, 0].long().tolist() for x, _ in DeviceBatchLoader(ds, 2 * B, shuffle=True, drop_last=True, device="cpu")]
    parts = []
    for r in range(2):
        torch.manual_seed(123)
        parts.append([x[:, 0].long().tolist() for x, _ in DeviceBatchLoader(ds, B, shuffle=True, drop_last=True, device="cpu", rank=r, world=2)])
    assert [a + b for a, b in zip(*parts)] == glob
    # ragged and empty shards
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    val = [x.shape[0] for x, _ in DeviceBatchLoader(ds, 25, shuffle=False, drop_last=False, device="cpu")]
    assert val == [25, 25, 25, 25, 3]
    # under data parallelism a ragged tail is trimmed to equal per-rank shares (103 = 2*50 + 3 -> ranks get 1 row each)
    for r in range(2):
        ld = DeviceBatchLoader(ds, 25, shuffle=False, drop_last=False, device="cpu", rank=r, world=2)
        sizes = [x.shape[0] for x, _ in ld]
        assert sizes == [25, 25, 1] and len(ld) == 3
    ld = DeviceBatchLoader(ds, 103, shuffle=False, drop_last=False, device="cpu", rank=1, world=4)
    assert [x.shape[0] for x, _ in ld] == [25] and len(ld) == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from behavio_temporal_vae_amd.parallel import GradSync, reduce_loglik, shard_range
    sync = GradSync()
    arena = torch.arange(20, dtype=torch.float32) * (rank + 1)      # "gradients": rank 0 -> i, rank 1 -> 2i
    enc_span = 8
    sync.launch(arena[enc_span:])       # decoder span first
    sync.launch(arena[:enc_span])
    sync.finish()
    scal = torch.tensor([1.0, 2.0, 3.0, 4.0]) * (rank + 1)
    sync.sum_(scal)
    b, e = shard_range(11, rank, world)
    ll = torch.arange(11, dtype=torch.float64)[b:e]
    tot, cnt = reduce_loglik(float(ll.sum()), ll.numel())
    # Monte-Carlo draws split over the ranks: each rank folds its K-shard of the reference fixture's log-weights into
    # a per-profile (max, sum of exp) state; two all-reduces must give the unsharded importance-weighted estimate
    import math
    from conftest import golden_spec, golden_state
    from oracle import refimpl as R
    from behavio_temporal_vae_amd.parallel import combine_lse
    g = load_golden("scoring_tiny")
    spec, _ = golden_spec(g)
    sd = golden_state(g)
    x, c, eps = (torch.from_numpy(np.array(g[k])) for k in ("x", "c", "eps"))
    K = eps.shape[0]
    with torch.no_grad():
        _, rlls, _ = R.iw_loglikelihood(sd, spec, x, c, eps)
        post = R.dist_forward(sd, "encoder", spec.enc, torch.cat((x, c), -1), False)
        z = post["mu"] + post["sigma"] * eps
        logw = rlls - R.normal_log_prob(z, post["mu"], post["sigma"]).sum(-1) + R.normal_log_prob(z, torch.zeros_like(z), torch.ones_like(z)).sum(-1)
    k0, k1 = shard_range(K, rank, world)
    mine = logw[k0:k1]
    m_run = mine.max(0).values if k1 > k0 else torch.full((x.shape[0],), -math.inf)
    s_run = torch.exp(mine - m_run).sum(0) if k1 > k0 else torch.zeros(x.shape[0])
    combine_lse(m_run, s_run)
    ll = m_run + torch.log(s_run) - math.log(K)
    # a rank without draws (K smaller than the world) contributes nothing
    m2 = torch.full((3,), -math.inf) if rank == 1 else torch.tensor([0.5, -2.0, 7.0])
    s2 = torch.zeros(3) if rank == 1 else torch.tensor([1.0, 2.0, 3.0])
    combine_lse(m2, s2)
    q.put((rank, arena.tolist(), scal.tolist(), tot, cnt, sync.world, ll.numpy(), m2.tolist(), s2.tolist()))
    dist.destroy_process_group()


def test_data_parallel_plumbing_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("scoring_tiny")
    for rank, arena, scal, tot, cnt, world, ll, m2, s2 in res:
        assert world == 2
        assert arena == [3.0 * i for i in range(20)]          # sum over ranks of (rank+1)*i
        assert scal == [3.0, 6.0, 9.0, 12.0]
        assert tot == float(sum(range(11))) and cnt == 11
        assert np.max(np.abs(ll - g["ll"]) / (np.abs(g["ll"]) + 1e-6)) < 1e-5      # K-sharded == the reference's ll
        assert m2 == [0.5, -2.0, 7.0] and s2 == [1.0, 2.0, 3.0]
