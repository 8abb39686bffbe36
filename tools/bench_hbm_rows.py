#!/usr/bin/env python3
"""SURVEY section 8(d) rows that are not MFMA-bound, in isolation (HIP-event timed):
  K18 VQ argmin   -- fp32 matrix pipe (exact): achieved fp32 TFLOP/s vs 157 peak, and HBM GB/s for completeness,
                     at the config size (6,144 rows = 96 frames) and at 2^20 rows
  K9  sequence assembly (embedding gathers + positional add) and K19 codebook gather -- HBM: achieved GB/s vs 8 TB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import ops

dev = 'cuda'


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


cb = (torch.randn(1024, 256, device=dev) * 0.5).contiguous()
ee = ops.vq_sqnorm(cb)
for rows in (6144, 1 << 20):
    z = torch.randn(rows, 256, device=dev)
    t = timeit(lambda: ops.vq_argmin(z, cb, ee), 20 if rows > 100000 else 200)
    fl, by = 2.0 * rows * 1024 * 256, rows * (256 * 4 + 8) + 1024 * 256 * 4
    print(f'K18 vq_argmin rows={rows:8d}: {t*1e6:9.1f} us  {fl/t/1e12:6.1f} TFLOP/s fp32 ({fl/t/1e12/157*100:4.1f} % of 157)  {by/t/1e9:7.1f} GB/s')

for B in (18, 288):
    L, E = 579, 768
    tabs = [torch.randn(5, E, device=dev), torch.randn(49408, E, device=dev), torch.randn(1026, E, device=dev), torch.randn(1026, E, device=dev)]
    seg = torch.tensor([0] + [1] * 64 + [0, 0] + [3] * 512, dtype=torch.int32, device=dev)
    ids = torch.stack([torch.randint(0, tabs[s].shape[0], (B, ), device=dev) for s in seg.tolist()], 1).contiguous()
    pos = torch.randn(L, E, device=dev)
    t = timeit(lambda: ops.assemble_sequence(tabs, ids, seg, pos))
    by = B * L * E * 4 * 2 + L * E * 4 + B * L * 8
    print(f'K9  assemble_sequence B={B:4d} L={L}: {t*1e6:9.1f} us  {by/t/1e9:7.1f} GB/s ({by/t/8e12*100:4.1f} % of 8 TB/s; rows read+written {by/1e6:.1f} MB)')

for n in (96, 4096):
    idx = torch.randint(0, 1024, (n, 64), device=dev)
    t = timeit(lambda: ops.gather_rows(cb, idx, torch.bfloat16))
    by = n * 64 * 256 * 2 + n * 64 * 8 + 1024 * 256 * 4  # rows written (bf16) + ids + the 1 MB table once (it stays in L2)
    print(f'K19 gather_rows frames={n:5d}: {t*1e6:9.1f} us  {by/t/1e9:7.1f} GB/s ({by/t/8e12*100:4.1f} % of 8 TB/s)')
