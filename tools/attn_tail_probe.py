#!/usr/bin/env python3
"""How much of the attention kernels' time is the partly filled last round?  Times forward / backward for batch sizes around the
resident-slot boundaries (L = 579, 12 heads: 60 blocks per sequence; 1,024 / 768 / 512 block slots for fwd / dQ / dK,dV)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev, bf = 'cuda', torch.bfloat16
L, H, E = 579, 12, 768
for B in (8, 9, 12, 13, 16, 17, 18):
    torch.manual_seed(0)
    qkv = (torch.randn(B * L, 3 * E, device=dev) * 0.5).to(bf)
    dO = (torch.randn(B * L, E, device=dev) * 0.1).to(bf)
    out = torch.empty(B * L, E, device=dev, dtype=bf)
    lse, delta = torch.empty(B * H * L, device=dev), torch.empty(B * H * L, device=dev)
    dqkv = torch.empty(B * L, 3 * E, device=dev, dtype=bf)
    st = ops._stream
    fwd = lambda: _lib.call('mmvid_attention_fwd', ops._p(qkv), 3 * E, B, L, H, E, 0.125, 2, 65, 65, 66, 66, ops._p(out), E, ops._p(lse), st())
    bwd = lambda: _lib.call('mmvid_attention_bwd', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E,
                            0.125, 2, 65, 65, 66, 66, ops._p(dqkv), 3 * E, st())
    fwd()
    tf, tb = timeit(fwd, 30), timeit(bwd, 30)
    print(f'B={B:2d} blocks {B * H * 5:5d}: fwd {tf * 1e3:6.1f} us ({tf * 1e3 / B:5.2f} per sequence) | bwd {tb * 1e3:6.1f} us ({tb * 1e3 / B:5.2f} per sequence)')
