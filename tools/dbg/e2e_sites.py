import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from behavio_temporal_vae_amd import engine as E, _native as nv
import behavio_temporal_vae_amd.vae_models as VM
dev = torch.device("cuda:0")
nv.ensure_device(dev)
models = []
orig = VM.CVAE.__init__
def patched(self, *a, **k):
    orig(self, *a, **k)
    models.append(self)
VM.CVAE.__init__ = patched
r = bench.e2e_benchmark(dev, 1000)
torch.cuda.synchronize()
f = E.GUARD.flags()
names = {}
for m in models:
    names.update({id(p): n for n, p in m.named_parameters()})
print("flagged:", [(names.get(k[0], "?"), k[1]) for k, s in E.GUARD.ids.items() if f[2 * s]])
