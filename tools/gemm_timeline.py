#!/usr/bin/env python3
"""Per-block timeline of the 256x128 GEMM (mmvid_gemm_trace): where a tile round's time goes, for gemm_epi 0 / 1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmvid_amd import _lib, ops

dev, bf = 'cuda', torch.bfloat16
M = 10422
for name, N, K, kw in (('qkv fwd', 2304, 768, {}), ('fc fwd', 3072, 768, {'gelu': True}), ('out fwd', 768, 768, {'res': True})):
    X = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
    bias = torch.zeros(N, device=dev)
    pre = torch.empty(M, N, device=dev, dtype=bf) if kw.get('gelu') else None
    res = torch.randn(M, N, device=dev) if kw.get('res') else None

    def run():
        if pre is not None:
            return ops.gemm(X, W, bias=bias, act=1, save_pre=pre)
        if res is not None:
            return ops.gemm(X, W, bias=bias, residual=res, out_dtype=torch.float32)
        return ops.gemm(X, W, bias=bias)

    for epi in (3, 4, 6):
        _lib.call('mmvid_set_option', b'gemm_epi', 1)
        _lib.call('mmvid_set_option', b'gemm_loader', 1 if epi >= 3 else 0)
        _lib.call('mmvid_set_option', b'gemm_debug', epi if epi in (4, 6) else 0)
        for _ in range(3):
            run()
        nblk = 256 if N > 768 else 246
        buf = torch.zeros(nblk * 2 * 8 * 8, device=dev, dtype=torch.int64)
        torch.cuda.synchronize()
        _lib.call('mmvid_gemm_trace', ops._p(buf))
        run()
        torch.cuda.synchronize()
        _lib.call('mmvid_gemm_trace', None)
        t = buf.cpu().numpy().reshape(nblk, 2, 8, 8).astype(np.float64) / 100.0  # us
        t0 = t[:, :, 0, 0][t[:, :, 0, 0] > 0].min()
        print(f'== {name} {M}x{N}x{K} gemm_epi {epi}: (us from the first block start; mean over blocks, leading wave group)')
        for tile in range(8):
            st = t[:, 0, tile]
            ok = st[:, 0] > 0
            if not ok.any():
                break
            s = st[ok] - t0
            print(f'   tile {tile}: {int(ok.sum()):3d} blocks  start {s[:,0].mean():6.2f}  first-K-visible +{(s[:,1]-s[:,0]).mean():5.2f}  '
                  f'K loop +{(s[:,2]-s[:,1]).mean():5.2f}  ' + (f'prologue-issue +{(s[:,3]-s[:,2]).mean():5.2f}  epilogue +{(s[:,4]-s[:,3]).mean():5.2f}' if epi else
                                                                f'epilogue +{(s[:,4]-s[:,2]).mean():5.2f}') + f'  end {s[:,4].mean():6.2f} (max {s[:,4].max():6.2f})'
                  + (f'  shader clock in the K loop {(100.0 * (st[ok][:,6]-st[ok][:,5]) / (st[ok][:,7]-st[ok][:,0])).mean():7.1f} MHz' if epi >= 3 else ''))
