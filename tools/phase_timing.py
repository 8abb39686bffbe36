#!/usr/bin/env python3
"""Diagnostic: wall time of each phase of the training step with a device sync between phases, next to the
host-only (enqueue) time of the same phase -- shows where the step is host-bound."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mmvid_amd.engine import FlatTrainer, backward_order

dev = torch.device('cuda', 0)
torch.manual_seed(0), np.random.seed(0)
model = bench.build_model(dev).train()
tr = FlatTrainer(model, order=backward_order)
gen = torch.Generator().manual_seed(0)
text, frames = bench.synth_batch(6, dev, gen)
for _ in range(2):
    bench.train_step(model, tr, text, frames)
torch.cuda.synchronize()


class T:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        torch.cuda.synchronize()
        self.t0 = time.perf_counter()
        return self

    def host(self):
        self.th = time.perf_counter() - self.t0

    def __exit__(self, *a):
        self.host()
        torch.cuda.synchronize()
        print(f'{self.name:28s} host {self.th*1e3:7.2f} ms   wall {1e3*(time.perf_counter()-self.t0):7.2f} ms')


for rep in range(2):
    print('--- rep', rep)
    with T('zero_grad'):
        tr.zero_grad()
    with T('encode target (48 frames)'):
        tok = model.get_image_tokens(frames)
    with T('warp + encode'):
        from mmvid_amd.dalle_bert import warp
        w = warp(frames, np.array([0.25] * 4))
        wtok = model.get_image_tokens(w)
    with T('full forward'):
        lm, lr, lv = model(text, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                           msm_strategy_prob=np.array([0.7, 0.1, 0.1, 0.1]), msm_bernoulli_prob=[0.2, 0.2])
        loss = 7 * lm + 0.5 * lr + 0.5 * lv
    with T('backward'):
        loss.backward()
    with T('optimizer step'):
        tr.step()
    with T('whole step (no inner syncs)'):
        bench.train_step(model, tr, text, frames)

# ---- finer: wrap the pieces of forward() (each wrapper syncs, so the sum is not the pipelined step time)
import mmvid_amd.dalle_bert as db

acc = {}


def wrap2(obj, name, label):
    fn = getattr(obj, name)

    def inner(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        tw = time.perf_counter() - t0
        e = acc.setdefault(label, [0.0, 0.0, 0])
        e[0] += th
        e[1] += tw
        e[2] += 1
        return r
    setattr(obj, name, inner)


wrap2(model, '_msm_mask', 'msm mask (host RNG + tiny kernels)')
wrap2(db, 'warp', 'warp (augmentation)')
wrap2(model, 'get_image_tokens', 'VQGAN encode (2*B*T frames)')
wrap2(model, '_assemble', 'assemble sequence')
wrap2(model, 'transformer_forward', 'tower forward')
wrap2(model, '_control_ids', 'control ids')
wrap2(model, '_small_head', 'small heads')
wrap2(db.LNLinearCrossEntropy, 'apply', 'to_logits + CE')
for rep in range(3):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lm, lr, lv = model(text, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                       msm_strategy_prob=np.array([0.7, 0.1, 0.1, 0.1]), msm_bernoulli_prob=[0.2, 0.2])
    loss = 7 * lm + 0.5 * lr + 0.5 * lv
    torch.cuda.synchronize()
    tf = time.perf_counter() - t0
    print(f'--- forward pieces, rep {rep}: forward total (with syncs) {tf*1e3:.2f} ms')
    for k, (th, tw, n) in acc.items():
        print(f'   {k:40s} calls {n}  host {th*1e3:7.2f} ms   wall {tw*1e3:7.2f} ms')
    print(f'   {"(rest of forward)":40s}          wall {(tf - sum(v[1] for v in acc.values()))*1e3:7.2f} ms')
    loss.backward()
    tr.step()
