#!/usr/bin/env python3
"""What the K loop of the 256x128 loader-wave GEMM block is made of: the loop as it is (gemm_debug 0), without a third of its LDS-DMA
requests (6), without any in-loop LDS-DMA (7), without its fragment reads (8), without both = the barrier + MFMA skeleton (9).  The
results of 6-9 are wrong by construction; only the stamped K-loop durations (mmvid_gemm_trace) are read."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mmvid_amd import _lib, ops

dev, bf = 'cuda', torch.bfloat16
M = 10422
for name, N, K, kmajor in (('qkv fwd NT', 2304, 768, False), ('d_h dX NN', 768, 2304, True), ('proj fwd NT', 768, 3072, False)):
    X = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(K, N, device=dev) * 0.03).to(bf) if kmajor else (torch.randn(N, K, device=dev) * 0.03).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    run = (lambda: ops.gemm(X, W, b_kmajor=kmajor, out=out))
    nt = (K + 63) // 64
    for stage, dbg in ((4, 0), (8, 0), (8, 7)) if not os.environ.get('ANATOMY') else ((4, 0), (4, 6), (4, 7), (4, 8), (4, 9), (4, 10), (4, 11)):
        _lib.call('mmvid_set_option', b'gemm_loaders', stage)
        _lib.call('mmvid_set_option', b'gemm_debug', dbg)
        for _ in range(3):
            run()
        nblk = 256
        buf = torch.zeros(nblk * 2 * 8 * 8, device=dev, dtype=torch.int64)
        torch.cuda.synchronize()
        _lib.call('mmvid_gemm_trace', ops._p(buf))
        run()
        torch.cuda.synchronize()
        _lib.call('mmvid_gemm_trace', None)
        t = buf.cpu().numpy().reshape(nblk, 2, 8, 8).astype(np.float64)
        rows = []
        for tile in range(8):
            st = t[:, 0, tile]
            ok = st[:, 0] > 0
            if not ok.any():
                break
            kl = (st[ok][:, 2] - st[ok][:, 1]) / 100.0
            mhz = (100.0 * (st[ok][:, 6] - st[ok][:, 5]) / (st[ok][:, 7] - st[ok][:, 0])).mean()
            rows.append(f'{kl.mean():6.2f} us = {kl.mean() * mhz / nt:6.0f} clk/K-tile @ {mhz:5.0f} MHz')
        print(f'{name:12s} {M}x{N}x{K} gemm_loaders {stage} gemm_debug {dbg}: K loop per tile: ' + ' | '.join(rows))
_lib.call('mmvid_set_option', b'gemm_debug', 0)
_lib.call('mmvid_set_option', b'gemm_loaders', 4)
