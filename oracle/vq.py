"""VQ lookup oracle: ctypes wrapper over vq_argmin.c + the reference's torch expression.

TEST INFRASTRUCTURE.  Reference: taming/modules/vqvae/quantize.py:297-341 (VectorQuantizer2.forward).
"""
import ctypes

import numpy as np
import torch

from .build import build

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_vq_argmin.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int64] * 3 + [ctypes.c_void_p] * 2
        _lib.oracle_vq_sqnorm.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    return _lib


def vq_argmin(z, e):
    """z [rows, dim] f32, e [n, dim] f32 -> (idx int64 [rows], dmin f32 [rows]); fixed fmaf-chain order."""
    z = np.ascontiguousarray(z.detach().cpu().numpy() if torch.is_tensor(z) else z, dtype=np.float32)
    e = np.ascontiguousarray(e.detach().cpu().numpy() if torch.is_tensor(e) else e, dtype=np.float32)
    rows, dim = z.shape
    n = e.shape[0]
    assert n <= 4096 and e.shape[1] == dim
    idx = np.empty(rows, dtype=np.int64)
    dmin = np.empty(rows, dtype=np.float32)
    _load().oracle_vq_argmin(z.ctypes.data, e.ctypes.data, rows, n, dim, idx.ctypes.data, dmin.ctypes.data)
    return torch.from_numpy(idx), torch.from_numpy(dmin)


def vq_distances_torch(z, e):
    """The reference's literal expression (quantize.py:306-308) in torch fp32."""
    return torch.sum(z**2, dim=1, keepdim=True) + torch.sum(e**2, dim=1) - 2 * (z @ e.t())
