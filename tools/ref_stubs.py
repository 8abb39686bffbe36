"""sys.modules stand-ins that let the *reference* (/root/reference) import in this container.

Used ONLY by tools/make_golden.py (golden-vector generation, build container only; the
reference never travels to the GPU box).  None of the reference's own arithmetic is
replaced: the stubs cover third-party packages that are not installed here
(SURVEY.md section 8c):

  torchvision                 -> RandomErasing restated from upstream (box sampling); io/utils/models empty
  axial_positional_embedding  -> AxialPositionalEmbedding restated from upstream (summed mode)
  pytorch_lightning           -> LightningModule = nn.Module
  omegaconf                   -> attr-dict over yaml.safe_load
  torch.jit.load              -> object whose state_dict() is a caller-supplied CLIP state_dict
"""
import math
import os
import sys
import types
from functools import reduce
from operator import mul

import torch
import yaml
from torch import nn

REF = '/root/reference'


# ----------------------------------------------------------------------------- torchvision
class RandomErasing(nn.Module):
    """Upstream torchvision.transforms.RandomErasing semantics (tensor input, value scalar)."""

    def __init__(self, p=0.5, scale=(0.02, 0.33), ratio=(0.3, 3.3), value=0, inplace=False):
        super().__init__()
        self.p, self.scale, self.ratio, self.value, self.inplace = p, scale, ratio, value, inplace

    @staticmethod
    def get_params(img, scale, ratio, value=None):
        img_c, img_h, img_w = img.shape[-3], img.shape[-2], img.shape[-1]
        area = img_h * img_w
        log_ratio = torch.log(torch.tensor(ratio))
        for _ in range(10):
            erase_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            h = int(round(math.sqrt(erase_area * aspect_ratio)))
            w = int(round(math.sqrt(erase_area / aspect_ratio)))
            if not (h < img_h and w < img_w):
                continue
            v = torch.tensor(value)[:, None, None]
            i = torch.randint(0, img_h - h + 1, size=(1, )).item()
            j = torch.randint(0, img_w - w + 1, size=(1, )).item()
            return i, j, h, w, v
        return 0, 0, img_h, img_w, img

    def forward(self, img):
        if torch.rand(1) < self.p:
            value = [float(self.value)]
            x, y, h, w, v = self.get_params(img, scale=self.scale, ratio=self.ratio, value=value)
            if not self.inplace:
                img = img.clone()
            img[..., x:x + h, y:y + w] = v.to(img.dtype) if torch.is_tensor(v) else v
            return img
        return img


# ------------------------------------------------------------- axial_positional_embedding
class AxialPositionalEmbedding(nn.Module):
    """Upstream lucidrains/axial-positional-embedding, summed mode (axial_dims=None)."""

    def __init__(self, dim, axial_shape, axial_dims=None):
        super().__init__()
        assert axial_dims is None
        self.dim, self.shape = dim, tuple(axial_shape)
        self.max_seq_len = reduce(mul, self.shape, 1)
        self.num_axials = len(self.shape)
        for ind, s in enumerate(self.shape):
            ax_shape = [1] * len(self.shape)
            ax_shape[ind] = s
            p = nn.Parameter(torch.zeros((1, *ax_shape, dim)).normal_(0, 1))
            setattr(self, f'weights_{ind}', p)

    def forward(self, x):
        b, t, e = x.shape
        embs = []
        for ind in range(self.num_axials):
            ax = getattr(self, f'weights_{ind}')
            embs.append(ax.expand((b, *self.shape, self.dim)).reshape(b, self.max_seq_len, self.dim))
        return sum(embs)[:, :t].to(x)


# ------------------------------------------------------------------------------ omegaconf
class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return _AttrDict({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


VQGAN_OVERRIDES = {}  # make_golden sets e.g. {'n_embed': 256, 'ch': 32} before building a VAE


class _OmegaConf:
    @staticmethod
    def load(path):
        with open(path) as f:
            cfg = _wrap(yaml.safe_load(f))
        p = cfg.model.params
        p['lossconfig'] = _AttrDict(target='torch.nn.Identity')  # real one downloads VGG weights
        if 'n_embed' in VQGAN_OVERRIDES:
            p['n_embed'] = VQGAN_OVERRIDES['n_embed']
        if 'ch' in VQGAN_OVERRIDES:
            p['ddconfig']['ch'] = VQGAN_OVERRIDES['ch']
        return cfg


# ------------------------------------------------------------------------------ install
CLIP_STATE = {}  # make_golden puts {'sd': <state_dict>} here before constructing BERT/DALLE


def install():
    sys.dont_write_bytecode = True
    os.chdir(REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)

    tv = types.ModuleType('torchvision')
    tv_t = types.ModuleType('torchvision.transforms')
    tv_t.RandomErasing = RandomErasing
    tv_io = types.ModuleType('torchvision.io')
    tv_io.write_video = lambda *a, **k: None
    tv_u = types.ModuleType('torchvision.utils')
    tv_m = types.ModuleType('torchvision.models')
    tv.transforms, tv.io, tv.utils, tv.models = tv_t, tv_io, tv_u, tv_m
    sys.modules.update({
        'torchvision': tv,
        'torchvision.transforms': tv_t,
        'torchvision.io': tv_io,
        'torchvision.utils': tv_u,
        'torchvision.models': tv_m,
    })

    ax = types.ModuleType('axial_positional_embedding')
    ax.AxialPositionalEmbedding = AxialPositionalEmbedding
    sys.modules['axial_positional_embedding'] = ax

    pl = types.ModuleType('pytorch_lightning')
    pl.LightningModule = nn.Module
    sys.modules['pytorch_lightning'] = pl

    oc = types.ModuleType('omegaconf')
    oc.OmegaConf = _OmegaConf
    sys.modules['omegaconf'] = oc

    class _Jit:
        def state_dict(self):
            return dict(CLIP_STATE['sd'])

    torch.jit.load = lambda *a, **k: _Jit()
