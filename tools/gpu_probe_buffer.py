#!/usr/bin/env python3
"""Probe buffer_load ... lds (LDS-DMA through a buffer descriptor) on gfx950: what lands in LDS for in-range and
out-of-range lanes, and whether soffset takes part in the range check -> gpurun_out/probe_buffer_lds.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

dev = 'cuda'
OOB = 0x7fffffff - 15


def run(voffs, soff, nrec):
    buf = torch.zeros(1024 + 1024, dtype=torch.int32)
    buf[:64] = torch.tensor(voffs, dtype=torch.int32)
    buf[64], buf[65] = soff, nrec
    buf[1024:] = torch.arange(1, 1025, dtype=torch.int32)
    buf = buf.to(dev)
    out = torch.zeros(256, dtype=torch.int32, device=dev)
    _lib.call('mmvid_probe', 1, ops._p(buf), ops._p(out), ops._stream())
    torch.cuda.synchronize()
    return out.cpu().view(64, 4)


with open('gpurun_out/probe_buffer_lds.txt', 'w') as f:
    def show(title, voffs, soff, nrec):
        r = run(voffs, soff, nrec)
        f.write(f'== {title} (soffset {soff}, num_records {nrec})\n')
        for l in (0, 1, 2, 3, 31, 32, 62, 63):
            exp = [(voffs[l] + soff) // 4 + 1 + e for e in range(4)]
            f.write(f'  lane {l:2d} voffset {voffs[l]:11d} -> {[hex(x & 0xffffffff) for x in r[l].tolist()]}  (in-range value would be {exp})\n')

    show('all in range', [l * 16 for l in range(64)], 0, 4096)
    show('odd lanes far out of range', [l * 16 if l % 2 == 0 else OOB for l in range(64)], 0, 4096)
    show('num_records 512: lanes >= 32 beyond it', [l * 16 for l in range(64)], 0, 512)
    show('soffset 256 with num_records 1024: does soffset count?', [l * 16 for l in range(64)], 256, 1024)
    show('soffset 2048, all in range', [l * 16 for l in range(64)], 2048, 4096)
    show('voffset -16 (0xfffffff0) + soffset 32', [0xfffffff0 - (1 << 32) if l == 0 else l * 16 for l in range(64)], 32, 4096)
print(open('gpurun_out/probe_buffer_lds.txt').read())
