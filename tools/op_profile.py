#!/usr/bin/env python3
"""Diagnostic: torch.profiler view of one training step -- which ATen ops (small copies, fills, elementwise
kernels issued by the Python glue) run besides the HIP library's launches, and from which source lines."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from mmvid_amd.engine import FlatTrainer, backward_order

dev = torch.device('cuda', 0)
torch.manual_seed(0), np.random.seed(0)
model = bench.build_model(dev).train()
tr = FlatTrainer(model, order=backward_order)
gen = torch.Generator().manual_seed(0)
text, frames = bench.synth_batch(6, dev, gen)
for _ in range(3):
    bench.train_step(model, tr, text, frames)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.train_step(model, tr, text, frames)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='count', row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by='count', row_limit=60, max_name_column_width=40, max_src_column_width=90))
