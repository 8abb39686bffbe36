#!/usr/bin/env python3
"""In-process A/B of SEVERAL configurations of the whole training step (tools/ab_graph.py compares the values of one knob).

    python tools/ab_multi.py base py:vae.stream=f32 py:vae.strict=split

Every argument is one configuration = the defaults + its comma-separated overrides (`base` = none).  The step is captured once per
configuration (engine.GraphedStep) and the graphs are replayed alternately in groups of 5 (GPU-bound timing, ~0.2 %; box-to-box and
run-to-run drift cancels)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mmvid_amd import _lib
from mmvid_amd.engine import FlatTrainer, GraphedStep, backward_order

DEFAULTS = {'py:vae.stream': 'bf16'}


def apply(model, name, v):
    if name.startswith('py:'):
        obj, path = model, name[3:].split('.')
        for n in path[:-1]:
            obj = getattr(obj, n)
        setattr(obj, path[-1], v)
    else:
        _lib.call('mmvid_set_option', name.encode(), int(v))


def main():
    configs = sys.argv[1:]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0), np.random.seed(0)
    model = bench.build_model(2, dev, 12).train()
    tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order)
    inputs = bench.synth_batch(6, 8, dev, torch.Generator().manual_seed(0))
    fn = bench.loss_fn(model, 2)
    for _ in range(2):
        bench.eager_step(tr, fn, inputs)
    steps = {}
    for c in configs:
        ov = {} if c == 'base' else dict(kv.split('=') for kv in c.split(','))
        for k, v in DEFAULTS.items():
            apply(model, k, ov.get(k, v))
        for k, v in ov.items():
            if k not in DEFAULTS:
                apply(model, k, v)
        steps[c] = GraphedStep(tr, fn, inputs, warmup=1)
    for k, v in DEFAULTS.items():
        apply(model, k, v)
    res = {c: [] for c in configs}
    for rep in range(8):
        for c in configs:
            steps[c]()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                steps[c]()
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / 5 * 1e3)
    base = np.median(res[configs[0]])
    for c in configs:
        m = np.median(res[c])
        print(f'{c:60s} median {m:7.3f} ms/step ({m - base:+.3f} vs {configs[0]})  groups {[round(x, 2) for x in res[c]]}')


if __name__ == '__main__':
    main()
