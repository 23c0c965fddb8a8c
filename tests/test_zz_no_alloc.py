"""The allocation-free column-sum entry point against the workspace-taking one it is a variant of.

``btvae_colsum_noalloc`` keeps its strip partials in a static device array inside the library; this test calls both
entry points through the C-ABI on the same inputs and requires identical results (same kernels, same partial layout)
on two calls in a row, the second on a smaller input, so that contents left in the shared array by the first call
would show up in the second."""
import ctypes

import pytest
import torch

from behavio_temporal_vae_amd import _native as nv

pytestmark = pytest.mark.gpu


def _both(G, accumulate, base):
    M, N = G.shape
    dev = G.device
    P = nv.ptr
    chunks = nv.lib().btvae_strip_chunks(M)
    ws = torch.empty(chunks * N, device=dev)
    out_a, out_b = base.clone(), base.clone()
    cur_a, cur_b = torch.empty(N, device=dev), torch.empty(N, device=dev)
    nv.call("btvae_colsum", P(G), P(out_a), P(cur_a), P(ws), M, N, int(accumulate))
    nv.call("btvae_colsum_noalloc", P(G), P(out_b), P(cur_b), M, N, int(accumulate))
    torch.cuda.synchronize()
    return (out_a, cur_a), (out_b, cur_b)


def test_colsum_noalloc_matches_colsum_twice_second_smaller():
    dev = torch.device("cuda:0")
    nv.ensure_device(dev)
    g = torch.Generator(device="cpu").manual_seed(7)
    # first call fills 256 chunks x 1000 columns of the static array; the second uses 3 chunks x 52 (scalar-column path)
    for (M, N), acc in (((65536, 1000), False), ((130, 52), True), ((4096, 48), False), ((1, 4), False)):
        G = torch.randn(M, N, generator=g).to(dev)
        base = torch.randn(N, generator=g).to(dev)
        (oa, ca), (ob, cb) = _both(G, acc, base)
        assert torch.equal(oa, ob), (M, N)
        assert torch.equal(ca, cb), (M, N)
        ref = G.double().sum(0)
        assert torch.allclose(cb.double(), ref, rtol=1e-5, atol=1e-3 * (M ** 0.5) * 1e-2 + 1e-5)
        want = ref + base.double() if acc else ref
        assert torch.allclose(ob.double(), want, rtol=1e-5, atol=1e-3 * (M ** 0.5) * 1e-2 + 1e-5)


def test_colsum_noalloc_makes_no_allocation_and_rejects_oversize():
    dev = torch.device("cuda:0")
    nv.ensure_device(dev)
    L = nv.lib()
    cap = L.btvae_colsum_noalloc_max_floats()
    assert cap == 1 << 20
    G = torch.randn(512, 64, device=dev)
    out = torch.empty(64, device=dev)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        nv.call("btvae_colsum_noalloc", nv.ptr(G), nv.ptr(out), None, 512, 64, 0)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0
    # 256 chunks x 4100 columns > 1 Mi floats: refused before any launch
    M, N = 65536, 4100
    rc = L.btvae_colsum_noalloc(ctypes.c_void_p(G.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, M, N, 0, nv.stream_ptr())
    assert rc != 0 and "exceed the static scratch" in nv.last_error()
