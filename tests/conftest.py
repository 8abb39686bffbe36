import json
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
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'))

    def __getitem__(self, k):
        return torch.from_numpy(self.z[k])

    def __contains__(self, k):
        return k in self.z.files

    def json(self, k):
        return json.loads(bytes(self.z[k]).decode())

    @property
    def meta(self):
        return self.json('meta')

    @property
    def manifest(self):
        return [(k, tuple(s)) for k, s in self.json('manifest')]


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def synth_model_sd(g, seed, vae_seed=11, cvae_seed=12):
    """state_dict of a golden case: synthetic weights keyed by the reference manifest."""
    from oracle.synth import synth_state_dict, synth_tensor
    sd = synth_state_dict(g.manifest, seed)
    for k in list(sd):
        if k.startswith('vae.'):
            sd[k] = synth_tensor(k[4:], sd[k].shape, vae_seed)
        elif k.startswith('cvae.'):
            sd[k] = synth_tensor(k[5:], sd[k].shape, cvae_seed)
    return sd


def relerr(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
