import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_state(g, prefix="sd::", dtype=torch.float32):
    sd = {}
    for k, v in g.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.array(v))
            sd[k[len(prefix):]] = t.to(dtype) if t.is_floating_point() else t
    return sd


def golden_spec(g):
    from oracle import refimpl as R
    cfg = json.loads(str(g["model_cfg"]))
    return R.model_spec(int(g["D"]), int(g["C"]) if "C" in g else 4, cfg), cfg


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-6))) if a.size else 0.0
