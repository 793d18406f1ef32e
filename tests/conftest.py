import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def dev():
    return torch.device('cuda:0')


@pytest.fixture(scope='session')
def det_sd():
    from drakes_amd.det_init import det_state_dict
    return det_state_dict(7)


@pytest.fixture(scope='session')
def gpu_model(dev, det_sd):
    import drakes_amd
    m = drakes_amd.Diffusion(drakes_amd.default_config()).to(dev)
    m.load_state_dict(det_sd, strict=True)
    m.eval()
    return m
