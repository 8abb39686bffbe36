"""Localise a non-finite value in the config-2 step: forward + backward replayed as a hipGraph, the optimiser step eager, with
the loss parts, the flat gradient and the parameters checked after every replay.
python tools/stress_nan2.py [steps] [seed] [graph|eager]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmvid_amd.engine import FlatTrainer, WarmupLR, backward_order  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 42
mode = sys.argv[3] if len(sys.argv) > 3 else 'graph'
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 2  # BASELINE config: 2 (text) or 4 (text + visual control)
dev = torch.device('cuda', 0)
torch.manual_seed(seed)
model = bench.build_model(cfg, dev, 12)
model.frontend.seed = seed
model.train()
tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order, lr_schedule=WarmupLR(1e-6, 1e-4, 5000, every=1))
gen = torch.Generator().manual_seed(seed)
batch = bench.synth_batch(6 if cfg == 2 else 2, 8, dev, gen, visuals=1 if cfg == 4 else 0)
vkw = dict(visual=batch['visual']) if cfg == 4 else {}
parts = torch.zeros(3, device=dev)
model._debug_keep = {}
model.transformer.debug_keep_saved = True


def fb():
    tr.zero_grad()
    lm, lr, lv = model(batch['text'], target=batch['frames'], return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                       msm_strategy_prob=bench.MSM_PROB, msm_bernoulli_prob=bench.MSM_BERN, vid_strategy_prob=bench.VID_PROB, **vkw)
    parts.copy_(torch.stack([lm.detach(), lr.detach(), lv.detach()]))
    loss = 7.0 * lm + 0.5 * lr + 0.5 * lv
    loss.backward()
    return loss.detach()


graph = None
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fb()
        tr.step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
if mode == 'graph':
    fe = model.frontend.step.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gl = fb()
    model.frontend.step.copy_(fe)
hist = []
for i in range(steps):
    if graph is not None:
        graph.replay()
        loss = gl
    else:
        loss = fb()
    v = float(loss)
    hist.append(v)
    gbad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in zip(tr.names, tr.params) if not torch.isfinite(p.grad).all()]
    if v != v or gbad:
        print(f'step {i}: loss {v} parts {parts.tolist()} non-finite grads in {len(gbad)} of {len(tr.params)} tensors: {gbad[:12]}')
        for k, t in model._debug_keep.items():
            if t is None:
                continue
            if t.is_floating_point():
                nb = int((~torch.isfinite(t)).sum())
                print(f'   {k}: {tuple(t.shape)} non-finite {nb}' + (f' rows {sorted(set((~torch.isfinite(t)).nonzero()[:, 0].tolist()))[:20]}' if nb and t.dim() > 1 else ''))
            else:
                print(f'   {k}: {tuple(t.shape)} min {int(t.min())} max {int(t.max())}')
        sv = getattr(model.transformer, '_last_saved', None)
        if sv is not None:
            nl = model.transformer.layers
            per = sv.numel() // nl
            M, E = model._debug_keep['y'].shape[0] * model._debug_keep['y'].shape[1], 768
            a256 = lambda x: (x + 255) // 256 * 256
            for li in range(nl):
                base = li * per
                xin = sv[base:base + M * E * 4].view(torch.float32).view(M, E)
                xmid = sv[base + a256(M * E * 4):base + a256(M * E * 4) + M * E * 4].view(torch.float32).view(M, E)
                bi, bm = ~torch.isfinite(xin), ~torch.isfinite(xmid)
                print(f'   layer {li}: x_in non-finite {int(bi.sum())} rows {sorted(set(bi.nonzero()[:, 0].tolist()))[:8]} | x_mid non-finite {int(bm.sum())} rows {sorted(set(bm.nonzero()[:, 0].tolist()))[:8]} cols {sorted(set(bm.nonzero()[:, 1].tolist()))[:8]}')
                if int(bm.sum()):
                    off = base + 2 * a256(M * E * 4) + 4 * a256(M * 4)
                    h1 = sv[off:off + M * E * 2].view(torch.bfloat16).view(M, E)
                    off += a256(M * E * 2)
                    qkv = sv[off:off + M * 3 * E * 2].view(torch.bfloat16).view(M, 3 * E)
                    off += a256(M * 3 * E * 2)
                    o = sv[off:off + M * E * 2].view(torch.bfloat16).view(M, E)
                    for nm, t in (('h1', h1), ('qkv', qkv), ('o', o)):
                        bb = ~torch.isfinite(t.float())
                        print(f'      {nm}: non-finite {int(bb.sum())} rows {sorted(set(bb.nonzero()[:, 0].tolist()))[:10]} cols {sorted(set(bb.nonzero()[:, 1].tolist()))[:10]}')
                    break
        break
    tr.step()
    pbad = [n for n, p in zip(tr.names, tr.params) if not torch.isfinite(p).all()]
    if pbad:
        print(f'step {i}: loss {v}, gradients finite, but parameters non-finite after the update: {len(pbad)} {pbad[:6]}')
        break
else:
    print(f'{mode}: {steps} steps, everything finite; last {hist[-2:]}')
