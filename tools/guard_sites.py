"""Which products trip the guard of the two-piece mode on the config0 training step (site = weight tensor + role):

    python tools/guard_sites.py [batch=4096] [iterations=30]
"""
import sys, os, contextlib, io
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from behavio_temporal_vae_amd.vae_models import CVAE
from behavio_temporal_vae_amd import engine as E
dev = torch.device("cuda:0")
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = CVAE(input_dim=bench.D, conditioner=bench.Cdim(bench.C), **bench.CONFIG0_MODEL)
model = model.to(dev).train()
model.train_kwargs = dict(bench.CONFIG0_TRAIN)
opt = model.get_optimizer(lr=1e-3, weight_decay=1e-5)
names = {id(p): n for n, p in model.named_parameters()}
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    x, c = bench.synth_batch(rng, B)
    model.train_core((torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)), opt)
    torch.cuda.synchronize()
    f = E.GUARD.flags()
    bad = [(names.get(k[0], k[0]), k[1]) for k, s in E.GUARD.ids.items() if f[2 * s]]
    if bad:
        print("iter", it, "flagged", bad, "fallbacks", E.GUARD.fallbacks)
        break
else:
    print("no site flagged in", it + 1, "iterations; fallbacks", E.GUARD.fallbacks)
