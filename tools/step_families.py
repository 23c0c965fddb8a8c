"""Per-family kernel time of the config0 training step (HIP events around every launch of 3 steps) and the dominant
launch of each product family:

    [DBG_B=4096] [BTVAE_STEP_GRAPH=0] python tools/step_families.py [fp32_split2 | fp32_split | fp32 | bf16]
"""
import sys, os, contextlib, io
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from behavio_temporal_vae_amd.vae_models import CVAE
from behavio_temporal_vae_amd import engine as E, _native as nv
dev = torch.device("cuda:0")
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = CVAE(input_dim=bench.D, conditioner=bench.Cdim(bench.C), **bench.CONFIG0_MODEL)
model = model.to(dev).train()
model.set_gemm_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
model.train_kwargs = dict(bench.CONFIG0_TRAIN)
opt = model.get_optimizer(lr=1e-3, weight_decay=1e-5)
rng = np.random.default_rng(0)
B = int(os.environ.get("DBG_B", "4096"))
x, c = bench.synth_batch(rng, B)
x, c = torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)
for it in range(5):
    model.train_core((x, c), opt)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for it in range(10):
    model.train_core((x, c), opt)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
nv.prof_filter(None, 0.0)
nv.prof_enable(True)
for it in range(3):
    model.train_core((x, c), opt)
torch.cuda.synchronize()
for fam in ("gemm_fwd", "gemm_dgrad", "gemm_wgrad"):
    ms, fl, n = nv.prof_dominant(fam)
    print(fam, "dominant", fl / 1e9, "GFLOP", ms / max(n, 1), "ms avg", n, "launches", fl * n / (ms * 1e-3) / 1e12 if ms else 0, "TF")
pr = nv.prof_collect()
nv.prof_enable(False)
tot = 0
for k, v in pr.items():
    print(k, round(v[0] / 3, 3), "ms/step", int(v[3] / 3), "launches", round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 and v[1] > 0 else "", "TF", round(v[2] / (v[0] * 1e-3) / 1e9) if v[0] > 0 else "", "GB/s")
    tot += v[0] / 3
print("sum", tot)
