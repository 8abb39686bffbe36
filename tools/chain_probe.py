#!/usr/bin/env python3
"""Floor of one phase of a persistent (single-launch) decode step: `mmvid_probe` 7 runs a chain of dependent phases over 256 co-resident
blocks, each phase = every block writes its share of a row, every block reads the whole row (tools/README.md).  mode 0: the words carry
a phase tag and are polled directly (data = flag); mode 1: a counter barrier, then the row is read."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

dev = 'cuda'
MODES = {0: 'tagged words', 1: 'counter barrier + loads', 2: 'tagged words, writers interleaved', 4: 'tagged words, one wave polls 12 words / lane',
         6: 'tagged words, interleaved writers, one wave polls'}
for blocks in (256, 128, 64, 32):
    for K in (768, 3072):
        for mode in (0, 1, 2, 4, 6):
            if K != 768 and mode >= 4:
                continue
            res = []
            for phases in (200, 1000):
                buf = torch.zeros(2 * K + 64, device=dev, dtype=torch.int64)
                arg = (ctypes.c_int32 * 4)(phases, blocks, K, mode)
                run = lambda: _lib.call('mmvid_probe', 7, ctypes.cast(arg, ctypes.c_void_p), ops._p(buf), ops._stream())
                buf.zero_()
                run()
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    buf.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                assert int(buf[2 * K + 1]) == 0, 'a spin ran out'
                res.append(min(ts))
            per = (res[1] - res[0]) / 800
            print(f'blocks {blocks} K {K} mode {MODES[mode]}: {per:6.2f} us per phase '
                  f'(200 phases {res[0]:8.1f} us, 1000 phases {res[1]:8.1f} us)')
