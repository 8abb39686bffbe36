"""HBM write rate of GEMM-epilogue-shaped store patterns (mmvid_probe which=2): is the epilogue bound by how it addresses
memory or by what else it does?  python tools/bench_store_pattern.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

names = ['256x128 tile, 8 B/lane, slab row order (the epilogue today)', '256x128 tile, 16 B/lane, natural row order',
         '256x128 tile, 8 B/lane, natural row order', '128x256 tile, 16 B/lane (512-B segments)', 'linear 64-KiB runs']
for M, N in ((10422, 2304), (10422, 3072), (10422, 768)):
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for variant in range(5):
        arg = (ctypes.c_int32 * 3)(M, N, variant)
        fn = lambda: _lib.call('mmvid_probe', 2, arg, ops._p(out), ops._stream())
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20
        print(f'[{M} x {N}] bf16 {M*N*2/1e6:6.1f} MB  {names[variant]:62s} {t*1e3:7.1f} us  {M*N*2/t/1e9:6.2f} TB/s')
