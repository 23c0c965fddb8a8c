import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from behavio_temporal_vae_amd import engine as E
dev = torch.device("cuda:0")

def report(tag):
    torch.cuda.synchronize()
    f = E.GUARD.flags()
    bad = [(k, s) for k, s in E.GUARD.ids.items() if f[2 * s]]
    print(tag, "flagged", len(bad), "fallbacks", E.GUARD.fallbacks, [k[1] for k, _ in bad][:12])
    E.GUARD.reset()

bench.scoring_benchmark(dev, 0, 1, 4096, 200, 100); report("scoring")
bench.imputation_benchmark(dev, 25000, 4); report("imputation")
bench.wide_benchmark(dev, 4096, 16, 3); report("wide")
bench.e2e_benchmark(dev, 300); report("e2e")
