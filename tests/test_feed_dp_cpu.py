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
    q.put((rank, arena.tolist(), scal.tolist(), tot, cnt, sync.world))
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
    for rank, arena, scal, tot, cnt, world in res:
        assert world == 2
        assert arena == [3.0 * i for i in range(20)]          # sum over ranks of (rank+1)*i
        assert scal == [3.0, 6.0, 9.0, 12.0]
        assert tot == float(sum(range(11))) and cnt == 11
