#!/usr/bin/env python3
"""Dump the lane layout of ds_read_b64_tr_b16 (gfx950) for a few address patterns -> gpurun_out/probe_tr.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

dev = 'cuda'
pats = {
    'lane*8': [l * 8 for l in range(64)],
    'all0': [0] * 64,
    'row-stride-136 (l&15)*136 + (l>>4)*8': [(l & 15) * 136 + (l >> 4) * 8 for l in range(64)],
    'lane*8 + 1024*(l>>4)': [l * 8 + 1024 * (l >> 4) for l in range(64)],
    '(l&3)*8 + (l>>2 &3)*256 + (l>>4)*2048': [(l & 3) * 8 + ((l >> 2) & 3) * 256 + (l >> 4) * 2048 for l in range(64)],
}
with open('gpurun_out/probe_tr.txt', 'w') as f:
    for name, offs in pats.items():
        o = torch.tensor(offs, dtype=torch.int32, device=dev)
        out = torch.zeros(64, 4, dtype=torch.int16, device=dev)
        _lib.call('mmvid_probe', 0, ops._p(o), ops._p(out), ops._stream())
        torch.cuda.synchronize()
        r = out.cpu().tolist()
        f.write(f'== {name}\n')
        for l in range(64):
            f.write(f'lane {l:2d} off {offs[l]:5d} (u16 idx {offs[l]//2:4d}) -> {r[l]}\n')
print(open('gpurun_out/probe_tr.txt').read()[:3000])
