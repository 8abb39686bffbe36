#!/usr/bin/env python3
"""The storer-wave GEMM block (option gemm_sw) against the round-3 loader-wave block on the step's multi-round GEMMs: HIP-event
times, and the per-tile timeline of the MFMA waves (mmvid_gemm_trace)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmvid_amd import _lib, ops

dev, bf = 'cuda', torch.bfloat16
M = int(os.environ.get('M', 10422))


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


cases = []
for name, N, K, kind in (('qkv fwd', 2304, 768, 'bias'), ('fc fwd', 3072, 768, 'gelu'), ('d_pre', 3072, 768, 'dact')):
    X = torch.randn(M, K, device=dev).to(bf)
    bias = torch.randn(N, device=dev)
    if kind == 'dact':
        W = (torch.randn(K, N, device=dev) * 0.03).to(bf)
        pre = torch.randn(M, N, device=dev).to(bf)
        cs = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=bf)
        run = (lambda X=X, W=W, pre=pre, cs=cs, out=out: ops.gemm(X, W, b_kmajor=True, dact_pre=pre, colsum=cs, out=out))
    elif kind == 'gelu':
        W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        pre = torch.empty(M, N, device=dev, dtype=bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        run = (lambda X=X, W=W, bias=bias, pre=pre, out=out: ops.gemm(X, W, bias=bias, act=1, save_pre=pre, out=out))
    else:
        W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        run = (lambda X=X, W=W, bias=bias, out=out: ops.gemm(X, W, bias=bias, out=out))
    cases.append((name, N, K, run))

for rep in range(2):
    for name, N, K, run in cases:
        fl = 2.0 * M * N * K
        row = f'{name:8s} {M}x{N}x{K}:'
        for sw in (0, 1):
            _lib.call('mmvid_set_option', b'gemm_sw', sw)
            t = timeit(run)
            row += f'  gemm_sw {sw}: {t:6.1f} us {fl / t / 1e6:7.1f} TF'
        print(row)
_lib.call('mmvid_set_option', b'gemm_sw', 1)
for name, N, K, run in cases:
    nblk = 256
    buf = torch.zeros(nblk * 2 * 8 * 8, device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    _lib.call('mmvid_gemm_trace', ops._p(buf))
    run()
    torch.cuda.synchronize()
    _lib.call('mmvid_gemm_trace', None)
    t = buf.cpu().numpy().reshape(nblk, 2, 8, 8).astype(np.float64) / 100.0  # us
    t0 = t[:, :, 0, 0][t[:, :, 0, 0] > 0].min()
    print(f'== {name} gemm_sw 1: per tile (us from the first block start; mean over blocks, leading wave group)')
    for tile in range(8):
        st = t[:, 0, tile]
        ok = st[:, 0] > 0
        if not ok.any():
            break
        s = st[ok] - t0
        print(f'   tile {tile}: {int(ok.sum()):3d} blocks  start {s[:,0].mean():6.2f}  first-K-visible +{(s[:,1]-s[:,0]).mean():5.2f}  '
              f'K loop +{(s[:,2]-s[:,1]).mean():5.2f}  convert+slab +{(s[:,3]-s[:,2]).mean():5.2f}  hand-off barrier +{(s[:,4]-s[:,3]).mean():5.2f}'
              f'  end {s[:,4].mean():6.2f} (max {s[:,4].max():6.2f})  shader clock {(100.0 * (st[ok][:,6]-st[ok][:,5]) / (st[ok][:,7]-st[ok][:,0])).mean():7.1f} MHz')
