#!/usr/bin/env python3
"""One GEMM / conv shape in a loop, for rocprofv3 --pmc passes (tools/gpu_pmc_gemm.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import ops

dev, bf = 'cuda', torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else 'fc'
M = 10422
if which == 'strip':
    x = torch.randn(54, 128, 128, 128, device=dev).to(bf)
    w = (torch.randn(128, 9, 128, device=dev) * 0.03).to(bf)
    b = torch.zeros(128, device=dev)
    fn = lambda: ops.conv3x3_strip(x, w, b)
elif which == 'conv':
    x = torch.randn(96, 128, 128, 128, device=dev).to(bf)
    w = (torch.randn(128, 9, 128, device=dev) * 0.03).to(bf)
    b = torch.zeros(128, device=dev)
    fn = lambda: ops.conv2d_nhwc(x, w, b, 0)
else:
    N, K = {'fc': (3072, 768), 'proj': (768, 3072), 'qkv': (2304, 768)}[which]
    X = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.03).to(bf)
    fn = lambda: ops.gemm(X, W)
for _ in range(10):
    fn()
torch.cuda.synchronize()
