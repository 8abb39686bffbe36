"""Where does a non-finite loss come from?  Runs bench.py's config-2 step eagerly and prints the three losses, the gradient
norm and the first non-finite gradient per step (python tools/debug_nan.py [B] [steps])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmvid_amd.engine import FlatTrainer, WarmupLR, backward_order  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device('cuda', 0)
torch.manual_seed(42)
model = bench.build_model(2, dev, 12)
model.frontend.seed = 42
model.train()
tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order, lr_schedule=WarmupLR(1e-6, 1e-4, 5000, every=1))
gen = torch.Generator().manual_seed(42)
batch = bench.synth_batch(B, 8, dev, gen)
for s in range(steps):
    tr.zero_grad()
    lm, lr, lv = model(batch['text'], target=batch['frames'], return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                       msm_strategy_prob=bench.MSM_PROB, msm_bernoulli_prob=bench.MSM_BERN, vid_strategy_prob=bench.VID_PROB)
    loss = 7.0 * lm + 0.5 * lr + 0.5 * lv
    loss.backward()
    bad = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in zip(tr.names, tr.params) if not torch.isfinite(p.grad).all()]
    gn = float(tr.G.double().pow(2).sum().sqrt())
    print(f'step {s}: msm {float(lm):.5f} rel {float(lr):.5f} vid {float(lv):.5f} |g| {gn:.4f} non-finite grads: {bad[:6]} ({len(bad)})', flush=True)
    tr.step()
    pbad = [n for n, p in zip(tr.names, tr.params) if not torch.isfinite(p).all()]
    if pbad:
        print('   non-finite parameters after the update:', pbad[:8], len(pbad))
        break
