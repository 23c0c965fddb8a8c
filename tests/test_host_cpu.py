"""CPU: host logic — the C-ABI library loads and exports every declared symbol, the model constructor
reproduces the reference's seeded initial weights and state-dict keys, and compute on CPU tensors fails loudly."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


class Cdim:
    def __init__(self, c):
        self.cond_dim = c


def test_library_exports_match_header():
    from behavio_temporal_vae_amd import _native as nv
    lib = nv.lib()
    hdr = open(os.path.join(ROOT, "include", "btvae_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(btvae_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found in include/btvae_hip.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(nv.EXPORTS) == declared, (set(nv.EXPORTS) ^ set(declared))
    assert lib.btvae_version() >= 100


def test_seeded_construction_matches_reference():
    from behavio_temporal_vae_amd.vae_models import CVAE
    g = load_golden("config0_smallbatch")
    cfg = json.loads(str(g["model_cfg"]))
    torch.manual_seed(int(g["seed"]))
    m = CVAE(input_dim=48, conditioner=Cdim(4), **cfg)
    sd = m.state_dict()
    assert len(sd) == 93
    for k, v in g.items():
        if k.startswith("init::"):
            assert np.array_equal(sd[k[6:]].numpy().reshape(-1)[:64], v), k
    assert m.encoder._num_parameters() == 5166096 and m.decoder._num_parameters() == 5218148
    assert set(m.model_kwargs) == {"latent_dim", "distribution_dict", "_"}
    assert abs(m.decoder.sigma_upper_lim - 0.6235382907247958) < 1e-12
    assert m.encoder.sigma_upper_lim == 1.0 and m.encoder.sigma_lower_lim == 0.05   # total_max_std is swallowed
    # parameters are views of one arena, gradients of another
    base = m._arena.data_ptr()
    assert all(base <= p.data_ptr() < base + m._arena.numel() * 4 for p in m.parameters())
    assert all(p.grad is not None for p in m.parameters())


def test_state_dict_keys_match_reference_fixture():
    from behavio_temporal_vae_amd.vae_models import CVAE
    g = load_golden("tiny_dict_2d")
    cfg = json.loads(str(g["model_cfg"]))
    m = CVAE(input_dim=int(g["D"]), conditioner=Cdim(int(g["C"])), **cfg)
    ref_keys = sorted(k[4:] for k in g if k.startswith("sd::"))
    assert sorted(m.state_dict().keys()) == ref_keys
    m.load_state_dict({k: torch.from_numpy(np.array(g["sd::" + k])) for k in ref_keys})
    assert np.array_equal(m.decoder.SigmaMapper.detach().numpy(), g["sd::decoder.SigmaMapper"])


def test_errors_and_no_cpu_fallback():
    from behavio_temporal_vae_amd.vae_models import CVAE
    from behavio_temporal_vae_amd.submodels import get_distribution_model, get_prior_params
    g = load_golden("tiny_dict_2d")
    cfg = json.loads(str(g["model_cfg"]))
    m = CVAE(input_dim=int(g["D"]), conditioner=Cdim(int(g["C"])), **cfg)
    with pytest.raises(RuntimeError):
        m.forward(torch.zeros(2, int(g["D"])), torch.zeros(2, int(g["C"])))
    with pytest.raises(ValueError):
        m.forward(torch.zeros(2, int(g["D"])), torch.zeros(2))
    with pytest.raises(NotImplementedError):
        get_distribution_model("laplace", input_dim=3, output_dim=2)
    with pytest.raises(ValueError):
        get_distribution_model("dict-gauss", input_dim=3, output_dim=8, vocab_size=8)
    with pytest.raises(NotImplementedError):
        get_prior_params("dict-gauss", 3)
    pp = get_prior_params(["normal", "g"], [2, 3])
    assert [tuple(p["mu"].shape) for p in pp] == [(2,), (3,)]
