import sys, os, contextlib, io, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from behavio_temporal_vae_amd import engine as E, _native as nv
dev = torch.device("cuda:0")
nv.ensure_device(dev)
# reuse the bench leg but profile every family
orig = bench.wide_benchmark
import types
src = bench.wide_benchmark
def run():
    nv.prof_filter(None, 0.0)
    r = src(dev, 4096, 16, 10)
    return r
r = run()
print(r["train"], r["validation"])
