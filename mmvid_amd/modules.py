"""Positional-embedding modules of the path.

AxialPositionalEmbedding restates the third-party `axial_positional_embedding` package the reference
imports (dalle_bert.py:8, 326-327; dalle_artv.py:141-146) in summed mode: parameters `weights_{i}` of shape
[1, 1.., s_i, ..1, dim] ~ N(0,1); forward(x[b,t,e]) = (sum_i broadcast(weights_i)).reshape(1, prod(shape), dim)[:, :t].
AxialPositionalEmbeddingList mirrors mmvid_pytorch/modules.py:8-53 (state_dict keys `module_list.{v}.weights_{i}`).
These are tiny [L, E] tables: plain torch ops (plumbing), their gradients flow through autograd."""
from functools import reduce
from operator import mul

import numpy as np
import torch
from torch import nn


class AxialPositionalEmbedding(nn.Module):
    def __init__(self, dim, axial_shape, axial_dims=None):
        super().__init__()
        assert axial_dims is None, 'only the summed mode is used on this path'
        self.dim, self.shape = dim, tuple(axial_shape)
        self.max_seq_len = reduce(mul, self.shape, 1)
        self.num_axials = len(self.shape)
        for ind, s in enumerate(self.shape):
            ax_shape = [1] * len(self.shape)
            ax_shape[ind] = s
            setattr(self, f'weights_{ind}', nn.Parameter(torch.zeros((1, *ax_shape, dim)).normal_(0, 1)))

    def table(self):
        """[max_seq_len, dim] dense table."""
        tot = 0
        for ind in range(self.num_axials):
            tot = tot + getattr(self, f'weights_{ind}').expand((1, *self.shape, self.dim))
        return tot.reshape(self.max_seq_len, self.dim)

    def forward(self, x):
        b, t, e = x.shape
        return self.table()[:t].unsqueeze(0).expand(b, t, e).to(x)


class AxialPositionalEmbeddingList(nn.Module):
    def __init__(self, dim=512, num=None, axial_shape=()):
        super().__init__()
        if num is None:
            num = axial_shape[0]
            axial_shape = axial_shape[1:]
        self.dim, self.num, self.axial_shape = dim, num, axial_shape
        self.chunk_size = int(np.prod(axial_shape))
        self.seq_len = num * self.chunk_size
        self.module_list = nn.ModuleList([AxialPositionalEmbedding(dim, axial_shape=axial_shape) for _ in range(num)])

    def table(self, insert_sep=False):
        """[num*chunk (+num if insert_sep), dim]; the [SEP] slot of each chunk gets zeros (modules.py:33-45)."""
        parts = []
        for m in self.module_list:
            parts.append(m.table())
            if insert_sep:
                parts.append(torch.zeros(1, self.dim, device=parts[-1].device, dtype=parts[-1].dtype))
        return torch.cat(parts, 0)

    def forward(self, emb):
        t = self.table(insert_sep=emb.shape[1] > self.seq_len)
        return t.unsqueeze(0).expand(emb.shape[0], -1, -1)
