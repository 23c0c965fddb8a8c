import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from behavio_temporal_vae_amd import engine as E
dev = torch.device("cuda:0")
from behavio_temporal_vae_amd import _native as nv
nv.ensure_device(dev)
for rep in (13, 15, 16, 17, 18, 20):
    nv.lib().btvae_guard_range_log2(rep)
    for on in (True,):
        E.GUARD.enabled = on
        E.GUARD.reset()
        r = bench.e2e_benchmark(dev, 1000)
        torch.cuda.synchronize()
        f = E.GUARD.flags()
        print("guard", on, "log2", rep, "e2e", round(r["value"]), "flagged", int((f[0::2] != 0).sum()), "fallbacks", E.GUARD.fallbacks, flush=True)
