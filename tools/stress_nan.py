"""Hunt for a non-finite loss in bench.py's config-2 loop: the same mix of graph replays and eager (per-launch timed) steps,
the loss checked after every step.  python tools/stress_nan.py [steps] [seed] [prof_every]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmvid_amd import _lib  # noqa: E402
from mmvid_amd.engine import FlatTrainer, GraphedStep, WarmupLR, backward_order  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 42
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda', 0)
torch.manual_seed(seed)
model = bench.build_model(2, dev, 12)
model.frontend.seed = seed
model.train()
tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order, lr_schedule=WarmupLR(1e-6, 1e-4, 5000, every=1))
gen = torch.Generator().manual_seed(seed)
batch = bench.synth_batch(6, 8, dev, gen)
parts = torch.zeros(3, device=dev)


def fn(text, frames):
    lm, lr, lv = model(text, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                       msm_strategy_prob=bench.MSM_PROB, msm_bernoulli_prob=bench.MSM_BERN, vid_strategy_prob=bench.VID_PROB)
    parts.copy_(torch.stack([lm.detach(), lr.detach(), lv.detach()]))
    return 7.0 * lm + 0.5 * lr + 0.5 * lv


bench.eager_step(tr, fn, batch)
g = GraphedStep(tr, fn, batch, warmup=2)
print('graph:', g.graph is not None, g.capture_error)
lib = _lib.load()
lib.mmvid_prof_begin(1)
lib.mmvid_prof_enable(0)
hist = []
for i in range(steps):
    eager = i % every == every - 1
    if eager:
        lib.mmvid_prof_enable(1)
        loss = bench.eager_step(tr, fn, batch)
        lib.mmvid_prof_enable(0)
    else:
        loss = g()
    v = float(loss)
    hist.append(v)
    if v != v or abs(v) == float('inf'):
        pbad = [n for n, p in zip(tr.names, tr.params) if not torch.isfinite(p).all()]
        gbad = [n for n, p in zip(tr.names, tr.params) if not torch.isfinite(p.grad).all()]
        print(f'step {i} ({"eager" if eager else "graph"}): loss {v}  parts {parts.tolist()}  non-finite params {len(pbad)} {pbad[:5]}  grads {len(gbad)} {gbad[:5]}')
        break
else:
    print(f'{steps} steps, all losses finite; last {hist[-3:]}')
print('loss history head', [round(h, 3) for h in hist[:12]])
