#!/usr/bin/env python3
"""Sustained (2 s per line) throughput of the c_fc- and c_proj-shaped GEMMs with parts switched off or changed: which part of a GEMM the
power-limited chip pays for.  gemm_debug 6 / 7: a third / all of the in-loop LDS-DMA skipped, 1: no epilogue stores (results wrong).  (Cache-policy bits on the
epilogue stores were tried here too: nt 62.7 -> 66.9 us, sc1 and sc0 sc1 65 -> 77 us on the c_fc shape: the default policy stays.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmvid_amd import _lib, ops
dev, bf = 'cuda', torch.bfloat16
def phase(name, fn, flops, seconds=2.0):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(50): fn()
        n += 50; torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'{name:72s} {dt/n*1e6:8.1f} us/call {flops*n/dt/1e12:8.1f} TFLOP/s', flush=True)
M = 10422
for N, K in ((3072, 768), (768, 3072)):
    X = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * 0.03).to(bf); out = torch.empty(M, N, device=dev, dtype=bf)
    X0 = torch.zeros_like(X); W0 = torch.zeros_like(W)
    fl = 2.0 * M * N * K
    phase(f'GEMM {M}x{N}x{K} random operands, as is', lambda: ops.gemm(X, W, out=out), fl)
    if True:
        for dbg, label in ((6, '2/3 of the in-loop DMA'), (7, 'no in-loop DMA (LDS contents static)'), (1, 'no epilogue stores')):
            _lib.call('mmvid_set_option', b'gemm_debug', dbg)
            phase(f'GEMM {M}x{N}x{K} random operands, {label}', lambda: ops.gemm(X, W, out=out), fl)
        _lib.call('mmvid_set_option', b'gemm_debug', 0)
        phase(f'GEMM {M}x{N}x{K} zero operands', lambda: ops.gemm(X0, W0, out=out), fl)
