import sys, os, contextlib, io, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from behavio_temporal_vae_amd.vae_models import CVAE
dev = torch.device("cuda:0")
os.environ["BTVAE_STEP_GRAPH"] = "0"
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    model = CVAE(input_dim=bench.D, conditioner=bench.Cdim(bench.C), **bench.CONFIG0_MODEL)
model = model.to(dev).train()
model.set_gemm_precision(prec)
model.train_kwargs = dict(bench.CONFIG0_TRAIN)
opt = model.get_optimizer(lr=1e-3, weight_decay=1e-5)
rng = np.random.default_rng(0)
x, c = bench.synth_batch(rng, 4096)
x, c = torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)
for it in range(6):
    model.train_core((x, c), opt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(10):
    model.train_core((x, c), opt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(18)
