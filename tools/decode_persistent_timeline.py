#!/usr/bin/env python3
"""Where a phase of the persistent decode step goes: shader-clock stamps (s_memtime, ~2.1 GHz) inside csrc/decode_persistent.hip for blocks 0, 1, 128, 255,
averaged over the layers of one step.  Segments: P1 poll | LN + dot + stores | P2 q poll | attention | P3 merge poll | dot + store |
P4 poll | LN + dot + stores | P5 poll + stage | dot + store."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from mmvid_amd.clip_tower import OpenAICLIPTransformer

dev = 'cuda'
L = 1152
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tw = OpenAICLIPTransformer(seq_len=L, which_model='openai_clip_visual', causal=True).to(dev).eval()
names = ['P1 poll', 'P1 LN,dot,store', 'P2 q poll', 'prefetch(+attention)', 'P3 merge poll', 'P3 dot,store', 'P4 poll', 'P4 LN,dot,store', 'P5 poll,stage',
         'P5 dot,store', 'next layer preload']
for P in (129, 1100):
    with torch.no_grad():
        cache = tw.new_kv_cache(B, L, dev)
        tw.prefill(torch.randn(B, P, 768, device=dev) * 0.5, cache)
        sess = tw.decode_session(cache, P, graph=False)
        assert sess.persistent
        xn = torch.randn(B, 768, device=dev) * 0.5
        for _ in range(3):
            sess.step(xn)
        buf = torch.zeros(4 * 12 * 16, dtype=torch.int64, device=dev)
        _lib.call('mmvid_decode_persistent_trace', ops._p(buf))
        sess.step(xn)
        torch.cuda.synchronize()
        _lib.call('mmvid_decode_persistent_trace', None)
    t = buf.view(4, 12, 16).cpu().double() / 2.1  # shader clocks -> ns at ~2.1 GHz
    print(f'batch {B}, position {P + 3}: step {(t[0, 11, 10] - t[0, 0, 0]) / 1e3:.1f} us (block 0, first stamp to last)')
    for bi, blk in enumerate((0, 1, 128, 255)):
        seg = []
        for i in range(11):
            nxt = t[bi, :, i + 1] if i < 10 else torch.cat([t[bi, 1:, 0], t[bi, 11:, 10]])
            d = (nxt - t[bi, :, i])
            if i == 10:
                d = d[:11]
            seg.append(d.mean().item())
        print(f'  block {blk:3d}: ' + ' | '.join(f'{n} {v:5.0f}' for n, v in zip(names, seg)) + f' | layer {sum(seg):6.0f} ns')
        if blk < 2:  # an attention unit: q in LDS -> (sync, new key) -> scores -> softmax statistics -> values -> reduce + store -> (prefetch) -> phase 3
            a = [t[bi, :, 3]] + [t[bi, :, i] for i in range(11, 16)] + [t[bi, :, 4]]
            print('             attention: ' + ' | '.join(f'{n} {(a[i + 1] - a[i]).mean().item():5.0f}' for i, n in enumerate(
                ['sync + new key', 'scores', 'softmax statistics', 'values', 'reduce + store', 'prefetch issue'])))
