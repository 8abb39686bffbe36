#!/usr/bin/env python3
"""K / M sweep of the forward GEMM to separate per-block fixed cost from per-K-tile cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmvid_amd import ops
from tools.bench_gemm import timeit
dev, bf = 'cuda', torch.bfloat16
for M in (8192, 32768):
    for N in (1024,):
        for K in (64, 128, 256, 512, 768, 1536, 3072, 6144):
            X = torch.randn(M, K, device=dev).to(bf); W = torch.randn(N, K, device=dev).to(bf)
            t = timeit(lambda: ops.gemm(X, W), 30)
            blocks = (M // 128) * (N // 128)
            print(f'M={M} N={N} K={K:5d} blocks={blocks:5d} rounds={blocks/512:5.2f}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:7.1f} TF  per-round {t*1e3/max(1,blocks/512):7.2f} us')
