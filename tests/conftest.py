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


from ab_opt_amd.utils.synth import AttrDict, build_model  # noqa: F401,E402  (kept importable from here for the tests)


@pytest.fixture(scope='session')
def model_factory():
    return build_model


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()
