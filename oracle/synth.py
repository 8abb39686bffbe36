"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Weights are a pure function of
(key name, shape, seed) through numpy's frozen legacy ``RandomState`` stream, so a
fixture only has to carry a seed and the key->shape manifest of the reference
state_dict, never the (tens of MB of) weights themselves.
"""
import zlib

import numpy as np
import torch


def _rs(seed, key):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)


def synth_tensor(key, shape, seed=0):
    """One tensor of the synthetic state_dict.

    Scale rules keep activations O(1) through deep stacks:
      * LayerNorm/GroupNorm gains (1-D '.weight' of a norm)      1 + 0.1 N(0,1)
      * biases                                                    0.02 N(0,1)
      * VQ codebook 'quantize.embedding.weight'                   0.5 N(0,1)  (well separated)
      * token / position embedding tables                         0.05 N(0,1) (axial weights_i too)
      * conv / linear / in_proj weights                           N(0,1)/sqrt(fan_in)
    """
    shape = tuple(int(s) for s in shape)
    r = _rs(seed, key)
    x = r.standard_normal(shape).astype(np.float32)
    last = key.rsplit('.', 1)[-1]
    if 'quantize.embedding' in key:
        x *= 0.5
    elif len(shape) == 1:
        if last == 'weight':
            x = 1.0 + 0.1 * x
        else:
            x *= 0.02
    elif ('emb' in key and 'embedding' not in key) or 'weights_' in last:
        x *= 0.05
    else:
        fan_in = int(np.prod(shape[1:]))
        x *= 1.0 / np.sqrt(max(fan_in, 1))
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


def synth_state_dict(manifest, seed=0):
    """manifest: iterable of (key, shape) -> {key: tensor}."""
    return {k: synth_tensor(k, s, seed) for k, s in manifest}


def synth_input(name, shape, seed=0, kind='normal'):
    r = _rs(seed, 'input:' + name)
    if kind == 'normal':
        return torch.from_numpy(r.standard_normal(tuple(shape)).astype(np.float32))
    if kind == 'uniform':
        return torch.from_numpy(r.random_sample(tuple(shape)).astype(np.float32))
    raise ValueError(kind)


def synth_tokens(name, shape, high, seed=0, low=0):
    r = _rs(seed, 'tok:' + name)
    return torch.from_numpy(r.randint(low, high, size=tuple(shape)).astype(np.int64))
