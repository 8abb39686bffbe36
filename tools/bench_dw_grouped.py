#!/usr/bin/env python3
"""Weight-gradient GEMMs of the tower backward, two ways (HIP-event timed):
  per layer   dW[N,K] += dY^T X with split-K through a workspace + the fixed-order reduce launch (what the layer loop does)
  grouped     the same product for ALL layers in one batched launch, no split-K: every block runs the full token reduction
Shapes: M = 10,422 tokens (18 sequences of 579), 12 layers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import ops

dev, bf = 'cuda', torch.bfloat16
M, LAYERS = 10422, int(os.environ.get('LAYERS', 12))


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


tot_a = tot_b = 0.0
for name, N, K in (('in_proj', 2304, 768), ('out', 768, 768), ('fc', 3072, 768), ('proj', 768, 3072)):
    dY = (torch.randn(LAYERS, M, N, device=dev) * 0.1).to(bf)
    X = (torch.randn(LAYERS, M, K, device=dev) * 0.1).to(bf)
    dW = torch.zeros(LAYERS, N, K, device=dev)
    out = torch.empty(LAYERS, N, K, device=dev)

    def per_layer():
        for l in range(LAYERS):
            ops.gemm_dw(dY[l], X[l], dW[l], accumulate=True)

    def grouped():
        ops.gemm(dY, X, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, out=out)

    ta, tb = timeit(per_layer), timeit(grouped)
    dW.zero_()
    per_layer()
    grouped()
    err = ((dW - out).abs().max() / out.abs().max()).item()
    fl = 2.0 * M * N * K * LAYERS
    print(f'dW {name:8s} {N}x{K}x{M} x{LAYERS}: per layer {ta:8.1f} us {fl/ta/1e6:7.1f} TF | grouped {tb:8.1f} us {fl/tb/1e6:7.1f} TF | '
          f'max diff / max {err:.2e}')
    tot_a += ta
    tot_b += tb
print(f'sum over the four kinds: per layer {tot_a/1e3:.3f} ms, grouped {tot_b/1e3:.3f} ms per backward')
