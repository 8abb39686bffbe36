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


def wide_vqgan_case(g):
    """(state_dict with 'model.' keys, frames [n,3,128,128]) of a tests/golden/vqgan_full16*.npz case (tools/make_golden.py::_vqgan_wide):
    synthetic weights of meta.seed; meta.codebook == 'reference_init' replaces the codebook rows by U(-1/n_e, 1/n_e), the reference's
    own initialisation (quantize.py:254), drawn from the seed."""
    from oracle.synth import synth_input, synth_state_dict
    meta = g.meta
    sd = synth_state_dict(g.manifest, meta['seed'])
    if meta['codebook'] == 'reference_init':
        k = 'model.quantize.embedding.weight'
        sd[k] = (synth_input('codebook_reference_init', tuple(sd[k].shape), meta['seed'], 'uniform') * 2 - 1) / sd[k].shape[0]
    img = synth_input('img_wide', (meta['n'], 3, meta['image_size'], meta['image_size']), meta['seed'], 'uniform')
    return sd, img


def flip_report(idx_mode, idx_ref, z_mode, z_ref, codebook, what=''):
    """Token-index disagreements of an encoder mode against the reference, with a bar that can fail (round 6; the round-5 form compared
    the gap with 64 x the LARGEST |dz| of the whole tensor, a bar above the median gap).  Per flipped token, with c = the mode's code
    and r = the reference's: gap = d(z_ref, c) - d(z_ref, r) (>= 0: r is the reference's argmin), and the measured |dd| = the error
    of exactly those two distances under the mode's z (|z|^2 is common to both and left out; fp64 from the fp32 rows).  A flip is
    legitimate only if gap <= 4 x |dd| (+ a few fp32 ulps of the distance: the reference expression itself rounds there).
    Returns (number of flips, rate, per-token gap / err of the reference's top-2 pair for histograms); `flip_report.last_gaps` holds
    the flipped tokens' reference gaps in units of the fp32 spacing of their distance (|z|^2 + |e|^2 - 2 z.e as the reference
    evaluates it): a gap of a few ulps is a tie at the reference's own resolution -- its z is not bit-reproducible across hosts."""
    zr, zm, e = z_ref.double().reshape(-1, z_ref.shape[-1]), z_mode.double().reshape(-1, z_ref.shape[-1]), codebook.double()
    im, ir = idx_mode.reshape(-1), idx_ref.reshape(-1)
    ee = (e * e).sum(1)
    Dr, Dm = ee[None, :] - 2.0 * zr @ e.t(), ee[None, :] - 2.0 * zm @ e.t()
    rows = torch.arange(zr.shape[0])
    fl = (im != ir).nonzero().view(-1)
    ulp = 8 * np.spacing(np.float32((zr * zr).sum(1).max().item() + ee.max().item()))
    flip_report.last_gaps = []
    for t in fl.tolist():
        c, r = int(im[t]), int(ir[t])
        gap = (Dr[t, c] - Dr[t, r]).item()
        flip_report.last_gaps.append(gap / float(np.spacing(np.float32(abs((zr[t] * zr[t]).sum().item() + Dr[t, r].item())))))
        dd = abs((Dm[t, c] - Dr[t, c]).item()) + abs((Dm[t, r] - Dr[t, r]).item())
        assert gap <= 4 * dd + ulp, f'{what}: token {t} flipped {r} -> {c} with a reference gap {gap:.3e} the measured |dd| {dd:.3e} does not explain'
    D2 = Dr.clone()
    D2[rows, ir] = float('inf')
    c2 = D2.argmin(1)
    gap_r = Dr[rows, c2] - Dr[rows, ir]
    err = ((Dm[rows, c2] - Dm[rows, ir]) - gap_r).abs().clamp_min(1e-30)
    return int(fl.numel()), fl.numel() / max(1, ir.numel()), (gap_r / err)
