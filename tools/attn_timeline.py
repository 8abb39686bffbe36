#!/usr/bin/env python3
"""Per-tile time line of the streaming attention forward kernel at the training shape (18 x 12 heads x L = 579): a first-round block
(4 waves per SIMD resident) and a tail-round block (nearly alone on its CU), from wall-clock stamps inside the kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

dev, bf = 'cuda', torch.bfloat16
B, L, H, E = 18, 579, 12, 768
torch.manual_seed(0)
qkv = (torch.randn(B * L, 3 * E, device=dev) * 0.5).to(bf)
out = torch.empty(B * L, E, device=dev, dtype=bf)
lse = torch.empty(B * H * L, device=dev)
trace = torch.zeros(1024 + 3 * 4096, device=dev, dtype=torch.int64)
dO = (torch.randn(B * L, E, device=dev) * 0.1).to(bf)
delta = torch.empty(B * H * L, device=dev)
dqkv = torch.empty(B * L, 3 * E, device=dev, dtype=bf)


def fwd():
    _lib.call('mmvid_attention_fwd', ops._p(qkv), 3 * E, B, L, H, E, 0.125, 2, 65, 65, 66, 66, ops._p(out), E, ops._p(lse), ops._stream())


def bwd():
    _lib.call('mmvid_attention_bwd', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E, 0.125, 2, 65,
              65, 66, 66, ops._p(dqkv), 3 * E, ops._stream())


for _ in range(3):
    fwd(), bwd()
_lib.call('mmvid_attention_trace', ops._p(trace))
fwd()
torch.cuda.synchronize()
bwd()
torch.cuda.synchronize()
_lib.call('mmvid_attention_trace', None)
full = trace.cpu().numpy().astype('int64')
t = full[:1024].reshape(2, 4, 16, 8)
names = ['loop top -> barrier passed', 'barrier -> S MFMAs issued (+ next DMA requests)', 'S issued -> softmax 0 done (P0 packed, V^T here)',
         'PV0 issue', 'PV0 issued -> softmax 1 done', 'PV1 issue']
for bi, bname in enumerate(('first-round block (4 waves per SIMD)', 'tail-round block (CU nearly idle)')):
    print(f'== {bname}; per wave: tile period and segment times in ns (100-MHz clock: 10-ns steps), tiles 1..8 averaged')
    for w in range(4):
        a = t[bi, w]
        if a[1, 0] == 0:
            print(f'  wave {w}: no stamps')
            continue
        period = (a[2:10, 0] - a[1:9, 0]).mean() * 10
        segs = [(a[1:9, i + 1] - a[1:9, i]).mean() * 10 for i in range(6)]
        print(f'  wave {w}: tile period {period:7.0f} ns | ' + ' | '.join(f'{s:6.0f}' for s in segs))
    print('   segments: ' + ' | '.join(names))
t0 = t[t > 0].min()
print('block start offsets (ns): first-round', (t[0, 0, 0, 0] - t0) * 10, ' tail-round', (t[1, 0, 0, 0] - t0) * 10, ' kernel span',
      (t.max() - t0) * 10)

import numpy as np
nb = B * H * 5
ent, ext = full[1024:1024 + 2 * nb:2], full[1025:1025 + 2 * nb:2]
e0 = ent.min()
print(f'block entry times (ns after the first): p10 {np.percentile(ent - e0, 10) * 10:.0f} p50 {np.percentile(ent - e0, 50) * 10:.0f} '
      f'p90 {np.percentile(ent - e0, 90) * 10:.0f} p95 {np.percentile(ent - e0, 95) * 10:.0f} max {(ent.max() - e0) * 10:.0f}')
print(f'block exit  times: p10 {np.percentile(ext - e0, 10) * 10:.0f} p50 {np.percentile(ext - e0, 50) * 10:.0f} p90 {np.percentile(ext - e0, 90) * 10:.0f} '
      f'max {(ext.max() - e0) * 10:.0f}')
dur = (ext - ent) * 10
print(f'block duration (wave 0: entry -> exit): p10 {np.percentile(dur, 10):.0f} p50 {np.percentile(dur, 50):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}')
print(f'traced first-round block: entry {(ent[100] - e0) * 10} first tile stamp {(t[0, 0, 0, 0] - e0) * 10} last tile end {(t[0, 0, 9, 6] - e0) * 10} exit {(ext[100] - e0) * 10}')
late = np.argsort(ent)[-60:]
print('latest-starting blocks:', sorted(late.tolist())[:12], '... entry', int((ent[late].min() - e0) * 10), '-', int((ent[late].max() - e0) * 10))

for name, base, tb in (('dQ kernel', 1024 + 4096, 256), ('dK/dV kernel', 1024 + 8192, 512)):
    ent, ext = full[base:base + 2 * nb:2], full[base + 1:base + 1 + 2 * nb:2]
    e0 = ent.min()
    dur = (ext - ent) * 10
    print(f'== {name}: block entry p50 {np.percentile(ent - e0, 50) * 10:.0f} p90 {np.percentile(ent - e0, 90) * 10:.0f} max {(ent.max() - e0) * 10:.0f} | '
          f'exit p10 {np.percentile(ext - e0, 10) * 10:.0f} p50 {np.percentile(ext - e0, 50) * 10:.0f} p90 {np.percentile(ext - e0, 90) * 10:.0f} '
          f'max {(ext.max() - e0) * 10:.0f} | duration p10 {np.percentile(dur, 10):.0f} p50 {np.percentile(dur, 50):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}')
    # duration by position of the block inside its (batch, head): 5 blocks each (xcd-remapped ids: group by id % 5 is only indicative)
    tt = full[tb:tb + 128].reshape(2, 4, 16)
    for bi, bn in enumerate(('first-round block', 'tail-round block')):
        per = [(tt[bi, w, 2:10] - tt[bi, w, 1:9]).mean() * 10 for w in range(4) if tt[bi, w, 1] > 0]
        print(f'   {bn}: tile period per wave (ns): ' + ' '.join(f'{x:.0f}' for x in per) + f' | entry {(ent[100 if bi == 0 else nb - 8] - e0) * 10} exit {(ext[100 if bi == 0 else nb - 8] - e0) * 10}')
    order = np.argsort(ent)
    print('   entry time of the blocks in dispatch order, every 64th:', ((ent[order][::64] - e0) * 10).tolist())

dc, dw = full[1002] - full[1000], (full[1003] - full[1001]) * 10
print(f'forward block 100: {dc} shader clocks in {dw} ns -> {dc / dw:.2f} GHz')
