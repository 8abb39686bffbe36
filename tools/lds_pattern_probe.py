#!/usr/bin/env python3
"""LDS read throughput of the attention kernels' access patterns on one CU (16 waves): bytes per shader clock."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

names = ['ds_read_b128, lane-contiguous (ideal)', 'ds_read_b128, K rows with chunk ^ ((row>>1)&7)  [attn.hip today]',
         'ds_read_b128, K rows with chunk ^ (row&7)', 'ds_read_b64_tr_b16 as tr_frag issues it', 'ds_read_b64, lane-contiguous (ideal)']
for pat in range(5):
    a = torch.tensor([pat], dtype=torch.int32, device='cuda')
    out = torch.zeros(8, dtype=torch.int64, device='cuda')
    for _ in range(2):
        _lib.call('mmvid_probe', 4, ops._p(a), ops._p(out), ops._stream())
    torch.cuda.synchronize()
    clk = out[:4].float().mean().item()
    nbytes = 16 * 2048 * 64 * (16 if pat <= 2 else 8)
    print(f'{names[pat]:70s}: {clk:9.0f} clocks per wave, {nbytes / clk:6.1f} B/clk per CU ({clk / 2048:5.1f} clocks per instruction and wave)')
