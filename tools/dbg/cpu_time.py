import sys, os, contextlib, io, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from behavio_temporal_vae_amd.vae_models import CVAE
dev = torch.device("cuda:0")
for prec in ("bf16", "fp32_split2"):
    for mode in ("0", "1"):
        os.environ["BTVAE_STEP_GRAPH"] = mode
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
        t0 = time.perf_counter()
        for it in range(20):
            model.train_core((x, c), opt)
        t_cpu = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(prec, "graph" if mode == "1" else "eager", "enqueue ms/step", round(t_cpu * 50, 2), "total ms/step", round(t_all * 50, 2), flush=True)
