import os
import sys
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope='session')
def golden():
    return load_golden


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


_MODELS = {}


def build_model(num_steps, seed, flavour='abdock', device='cpu'):
    """Product model with hash-filled weights (same fill as the reference got in make_golden.py)."""
    import cases
    from ab_opt_amd import get_model
    from ab_opt_amd.utils import synth
    key = (num_steps, seed, flavour, str(device))
    if key not in _MODELS:
        cfg = cases.cfg_abdock(num_steps)
        if flavour == 'abdesign':
            for k in ('num_bins', 'dist_min', 'dist_max'):
                cfg.pop(k)
            cfg['diffusion'].pop('obj')
        m = get_model(AttrDict(cfg)).eval()
        synth.fill_module_(m, seed=seed)
        _MODELS[key] = m.to(device)
    return _MODELS[key]


@pytest.fixture(scope='session')
def model_factory():
    return build_model


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()
