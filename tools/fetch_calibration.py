#!/usr/bin/env python3
"""Known-size read streams for calibrating rocprofv3's FETCH_SIZE (run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`):
1 GiB through buffer_load_dwordx4 ... lds, 1 GiB through global_load_dwordx4, and a torch copy of 1 GiB (reads 1 GiB, writes 1 GiB)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

n = 1 << 30
buf = torch.empty(n, device='cuda', dtype=torch.uint8).random_(0, 255)
dst = torch.empty_like(buf)
for mode in (0, 1):
    arr = (ctypes.c_int64 * 2)(n, mode)
    for _ in range(2):
        _lib.call('mmvid_probe', 6, arr, ops._p(buf), ops._stream())
torch.cuda.synchronize()
for _ in range(2):
    dst.copy_(buf)
torch.cuda.synchronize()
print('streams done: probe_stream_kernel dispatches 1-2 = LDS-DMA, 3-4 = global_load, then the copies; each reads', n, 'bytes')
