#!/usr/bin/env python3
"""Anatomy of the two GEMM epilogues: full / no stores (gemm_debug 1) / no K loop (gemm_debug 2) for gemm_epi 0 and 1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev, bf = 'cuda', torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10422
for name, N, K, kw in (('qkv fwd', 2304, 768, {}), ('fc fwd (+gelu, pre saved)', 3072, 768, {'gelu': True}), ('out fwd f32+res', 768, 768, {'res': True}),
                       ('proj fwd f32+res', 768, 3072, {'res': True})):
    X = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
    bias = torch.zeros(N, device=dev)
    pre = torch.empty(M, N, device=dev, dtype=bf) if kw.get('gelu') else None
    res = torch.randn(M, N, device=dev) if kw.get('res') else None
    fl = 2.0 * M * N * K
    row = f'{name:28s} {M}x{N}x{K}:'
    for epi in (0, 1):
        _lib.call('mmvid_set_option', b'gemm_epi', epi)
        row += f' | epi {epi}:'
        for dbg, label in ((0, 'full'), (1, 'no stores'), (2, 'no K loop')):
            _lib.call('mmvid_set_option', b'gemm_debug', dbg)
            if pre is not None:
                t = timeit(lambda: ops.gemm(X, W, bias=bias, act=1, save_pre=pre))
            elif res is not None:
                t = timeit(lambda: ops.gemm(X, W, bias=bias, residual=res, out_dtype=torch.float32))
            else:
                t = timeit(lambda: ops.gemm(X, W, bias=bias))
            row += f' {label} {t*1e3:6.1f}'
    _lib.call('mmvid_set_option', b'gemm_debug', 0)
    print(row, flush=True)
