#!/usr/bin/env python3
"""LayerNorm forward / backward at the tower's shape ([10422, 768]): time and achieved HBM rate against the grid cap of the
backward's two-stage reduction (the workspace size decides it)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev = 'cuda'
M, E = 10422, 768
torch.manual_seed(0)
x = torch.randn(M, E, device=dev)
w, b = torch.randn(E, device=dev), torch.randn(E, device=dev)
t = timeit(lambda: ops.layernorm_fwd(x, w, b), 50)
print(f'LN fwd: {t*1e3:6.1f} us  {(M*E*6 + M*8)/t/1e9:6.2f} TB/s (algorithmic: 4 B read + 2 B written per element)')
y, mean, rstd = ops.layernorm_fwd(x, w, b)
dy16 = torch.randn(M, E, device=dev).bfloat16()
g = torch.randn(M, E, device=dev)
gb = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
dw, db, dc = (torch.zeros(E, device=dev) for _ in range(3))
for cap in (128, 256, 512, 1024, 2048):
    ws = torch.empty(cap * 3 * E, device=dev)

    def run():
        _lib.call('mmvid_layernorm_bwd_ex', ops._p(dy16), 1, E, ops._p(x), E, ops._p(mean), ops._p(rstd), ops._p(w), M, E, ops._p(g), E, 1,
                  ops._p(gb), ops._p(dw), ops._p(db), ops._p(dc), ops._p(ws), ws.numel(), ops._stream())

    t = timeit(run, 50)
    alg = M * E * (2 + 4 + 4 + 4 + 2)
    print(f'LN bwd (bf16 dy, add, bf16 copy, dw/db/colsum) grid cap {cap:5d}: {t*1e3:6.1f} us  {alg/t/1e9:6.2f} TB/s algorithmic '
          f'(+ {cap*3*E*4*2/1e6:.1f} MB of partial rows written and re-read)')
