#!/usr/bin/env python3
"""Attention forward / backward at the training step's shape (18 sequences x 12 heads x L = 579, head_dim 64): time and TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev, bf = 'cuda', torch.bfloat16
B, L, H, E = 18, 579, 12, 768
torch.manual_seed(0)
qkv = (torch.randn(B * L, 3 * E, device=dev) * 0.5).to(bf)
dO = (torch.randn(B * L, E, device=dev) * 0.1).to(bf)
out = torch.empty(B * L, E, device=dev, dtype=bf)
lse = torch.empty(B * H * L, device=dev)
delta = torch.empty(B * H * L, device=dev)
dqkv = torch.empty(B * L, 3 * E, device=dev, dtype=bf)
st = ops._stream


def fwd():
    _lib.call('mmvid_attention_fwd', ops._p(qkv), 3 * E, B, L, H, E, 0.125, 2, 65, 65, 66, 66, ops._p(out), E, ops._p(lse), st())


def bwd():
    _lib.call('mmvid_attention_bwd', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E, 0.125, 2, 65,
              65, 66, 66, ops._p(dqkv), 3 * E, st())


nws = _lib.load().mmvid_attention_bwd_workspace_bytes(B, L, H)
ws = torch.empty(nws, device=dev, dtype=torch.uint8)


def bwd_ws():  # the tower's call: with the workspace that lets the last round's blocks be split (option attn_tail)
    _lib.call('mmvid_attention_bwd_ws', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E, 0.125, 2, 65,
              65, 66, 66, ops._p(dqkv), 3 * E, None, ops._p(ws), nws, st())


fwd()
tf, tb, tw = timeit(fwd, 30), timeit(bwd, 30), timeit(bwd_ws, 30)
print(f'bwd with the tail-split workspace {tw*1e3:6.1f} us (without {tb*1e3:6.1f})')
fl = 4.0 * B * H * L * L * 64
print(f'attention fwd {tf*1e3:6.1f} us {fl/tf/1e9:6.1f} TF | bwd (dQ + dK/dV) {tb*1e3:6.1f} us {2.5*fl/tb/1e9:6.1f} TF')
# reference check against torch (fp32 math on the bf16 inputs)
q, k, v = [t.float().view(B, L, H, 64).transpose(1, 2) for t in qkv.float().split(E, dim=1)]
mask = torch.zeros(L, L, device=dev)
mask[65, :65] = float('-inf')
mask[66, :66] = float('-inf')
q.requires_grad_(True), k.requires_grad_(True), v.requires_grad_(True)
o = torch.softmax(q @ k.transpose(-1, -2) * 0.125 + mask, -1) @ v
o2 = o.transpose(1, 2).reshape(B * L, E)
o2.backward(dO.float())
err = lambda a, b: ((a.float() - b).abs().max() / b.abs().max()).item()
g = torch.cat([t.grad.transpose(1, 2).reshape(B * L, E) for t in (q, k, v)], 1)
bwd_ws()
print(f'rel err with the tail split: dqkv {err(dqkv, g):.2e}')
assert err(dqkv, g) < 3e-2
bwd()
db = torch.zeros(3 * E, device=dev)
_lib.call('mmvid_attention_bwd_bias', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E, 0.125, 2, 65,
          65, 66, 66, ops._p(dqkv), 3 * E, ops._p(db), st())
print(f'rel err: out {err(out, o2):.2e} dqkv {err(dqkv, g):.2e} fused in_proj bias gradient {err(db, g.sum(0)):.2e}')
assert err(db, g.sum(0)) < 1e-2
assert err(out, o2) < 2e-2 and err(dqkv, g) < 3e-2
