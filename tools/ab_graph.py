#!/usr/bin/env python3
"""In-process A/B of a library tuning knob on the whole training step.

    python tools/ab_graph.py <option> <value> <value> [...]      e.g.  py:vae.stream f32 bf16   |   py:vae.strict mixed split   |   graphs 0 1

The step is captured once per value (engine.GraphedStep; the knob is baked into the capture) and the graphs are
replayed alternately in groups of 5: GPU-bound timing, stable to ~0.2 %, and box-to-box / run-to-run drift cancels
(eager steps on this pool fluctuate by tens of percent with the host)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mmvid_amd import _lib
from mmvid_amd.engine import FlatTrainer, GraphedStep, backward_order


def main():
    opt = sys.argv[1]
    values = [v if opt.startswith('py:') and not v.lstrip('-').isdigit() else int(v) for v in sys.argv[2:]]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0), np.random.seed(0)
    model = bench.build_model(2, dev, 12).train()
    tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order)
    inputs = bench.synth_batch(6, 8, dev, torch.Generator().manual_seed(0))
    fn = bench.loss_fn(model, 2)
    for _ in range(2):
        bench.eager_step(tr, fn, inputs)
    steps = {}
    for v in values:
        if opt.startswith('py:'):  # a Python attribute of the model instead of a library option, e.g. py:vae.streams 1 2
            obj, path = model, opt[3:].split('.')
            for name in path[:-1]:
                obj = getattr(obj, name)
            setattr(obj, path[-1], v)
        else:
            _lib.call('mmvid_set_option', opt.encode(), v)
        steps[v] = GraphedStep(tr, fn, inputs, warmup=1)
    res = {v: [] for v in values}
    for rep in range(8):
        for v in values:
            steps[v]()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                steps[v]()
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / 5 * 1e3)
    for v in values:  # host cost of ONE replay issued into an idle queue (no back-pressure from earlier work)
        hs = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps[v]()
            hs.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
        print(f'{opt}={v}: host time of one replay call (copies + hipGraphLaunch) {[round(x, 2) for x in hs]} ms')
    for v in values:
        print(f'{opt}={v}: ms/step per group {[round(x, 2) for x in res[v]]}  median {np.median(res[v]):.2f}')


if __name__ == '__main__':
    main()
