#!/usr/bin/env python3
"""Can an HBM-bound kernel hide beside a matrix-pipe-bound one?  VQGAN encode of 54 frames (strip convolutions: 8 waves x ~200 registers and
144 KiB of LDS per CU) and the fused clip + Adam update of 125 M parameters (no LDS, ~40 registers), one after the other on one stream
against side by side on two streams -- what a step that applies the previous step's update while it tokenises its own batch could gain."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mmvid_amd.engine import FlatTrainer, backward_order

dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = bench.build_model(2, dev, 12).train()
tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order)
frames = torch.rand(54, 3, 128, 128, device=dev)
tr.G.normal_()
side = torch.cuda.Stream()


def enc():
    return model.vae.get_codebook_indices(frames)


def upd():
    tr.G.mul_(1.0)  # (keeps G non-zero; cheap)
    tr.step()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def both_serial():
    enc()
    upd()


def both_parallel():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        upd()
    enc()
    cur.wait_stream(side)


a, b = timed(enc), timed(upd)
c, d = timed(both_serial), timed(both_parallel)
print(f'VQGAN encode of 54 frames {a:.3f} ms | clip + Adam {b:.3f} ms | one after the other {c:.3f} ms | on two streams {d:.3f} ms')
