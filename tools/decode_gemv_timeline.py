#!/usr/bin/env python3
"""Inside one decode gemv (csrc/decode.hip): wall-clock stamps of every block -- entry, loads issued, input rows arrived, LayerNorm
done, rows staged in LDS, weights arrived, exit -- for the four launches of a layer at the ART-V width."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mmvid_amd import _lib, ops

dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
m = bench.build_model(5, dev).eval()
tw = m.transformer
E, F = 768, 3072
x = torch.randn(B, E, device=dev)
xf = torch.randn(B, F, device=dev)
sh = tw._sync_shadow()
blk = tw.transformer.resblocks[0]
ln = (blk.ln_1.weight, blk.ln_1.bias, 1e-5)
kinds = [('LN+qkv  768->2304', lambda: ops.gemv_rows(x, sh[0], blk.attn.in_proj_bias, ln=ln, round_in=True), 288),
         ('out     768->768 ', lambda: ops.gemv_rows(x.bfloat16().float(), sh[1], blk.attn.out_proj.bias, residual=x, round_in=True), 192),
         ('LN+fc   768->3072', lambda: ops.gemv_rows(x, sh[2], blk.mlp.c_fc.bias, ln=ln, act=1, round_in=True), 384),
         ('proj   3072->768 ', lambda: ops.gemv_rows(xf, sh[3], blk.mlp.c_proj.bias, residual=x, round_in=True), 192)]
trace = torch.zeros(512 * 8, device=dev, dtype=torch.int64)
names = ['entry', 'loads issued', 'rows arrived', 'LayerNorm done', 'rows in LDS', 'weights arrived', 'exit']
print(f'batch {B}; times in ns after the first block entered (p50 / p90 / max over the blocks)')
for name, fn, nb in kinds:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    # a preceding dependent launch, as in the real chain (the weights are cold in L2: 170 MB are streamed between two uses)
    trace.zero_()
    _lib.call('mmvid_decode_trace', ops._p(trace))
    fn()
    torch.cuda.synchronize()
    _lib.call('mmvid_decode_trace', None)
    t = trace.cpu().numpy().reshape(512, 8)[:nb, :7].astype('int64')
    t0 = t[:, 0].min()
    row = []
    for i, nm in enumerate(names):
        v = (t[:, i] - t0) * 10
        v = v[t[:, i] > 0]
        row.append(f'{nm} {np.percentile(v, 50):.0f}/{np.percentile(v, 90):.0f}/{v.max():.0f}')
    print(f'{name} ({nb} blocks): ' + ' | '.join(row))
