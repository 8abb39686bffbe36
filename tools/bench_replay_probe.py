#!/usr/bin/env python3
"""Why bench.py's headline (first replays right after the capture) sits ~1 ms above tools/ab_graph.py's steady-state replays of the
same captured step: bench.py's own set-up, then 4 batches of 20 replays timed as bench.py times them, then 20 replays with a sync after
each (per-step distribution), and the host time of each replay call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random

import numpy as np
import torch

import bench
from mmvid_amd import _lib
from mmvid_amd.engine import FlatTrainer, GraphedStep, WarmupLR, backward_order

dev = torch.device('cuda', 0)
torch.set_num_threads(4)
seed = 42
random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
model = bench.build_model(2, dev, 12)
model.frontend.seed = seed
model.train()
tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order, lr_schedule=WarmupLR(1e-6, 1e-4, 5000, every=1))
batch = bench.synth_batch(6, 8, dev, torch.Generator().manual_seed(seed))
fn = bench.loss_fn(model, 2)
bench.eager_step(tr, fn, batch)
g = GraphedStep(tr, fn, batch, warmup=2)
lib = _lib.load()
if len(sys.argv) > 1 and sys.argv[1] == 'prof':
    lib.mmvid_prof_begin(1)
    lib.mmvid_prof_enable(0)
torch.cuda.synchronize()
for b in range(4):
    t0 = time.perf_counter()
    hs = []
    for _ in range(20):
        th = time.perf_counter()
        g()
        hs.append((time.perf_counter() - th) * 1e3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20 * 1e3
    print(f'batch {b}: {dt:.3f} ms/step over 20 replays; host call ms: first {hs[0]:.2f}, median {np.median(hs):.2f}, max {max(hs):.2f}')
per = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g()
    torch.cuda.synchronize()
    per.append((time.perf_counter() - t0) * 1e3)
print('one replay at a time (sync after each):', [round(x, 2) for x in per])
# the same graph timed with events around 10 replays
a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    g()
z.record()
torch.cuda.synchronize()
print(f'events around 10 replays: {a.elapsed_time(z) / 10:.3f} ms/step')
