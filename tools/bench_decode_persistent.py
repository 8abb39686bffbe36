#!/usr/bin/env python3
"""One ART-V decode step through the 12-layer tower: the five-launches-per-layer form against the persistent single-launch form
(csrc/decode_persistent.hip), both as hipGraph replays, at short and long cache positions."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd.clip_tower import OpenAICLIPTransformer

dev = 'cuda'
L = 1152
tw = OpenAICLIPTransformer(seq_len=L, which_model='openai_clip_visual', causal=True).to(dev).eval()
for B in [int(v) for v in (sys.argv[1:] or ['1', '2', '4'])]:
    for P in (129, 1100):
        x = torch.randn(B, P, 768, device=dev) * 0.5
        row = []
        with torch.no_grad():
            for form in ('launches', True):
                cache = tw.new_kv_cache(B, L, dev)
                tw.prefill(x, cache)
                sess = tw.decode_session(cache, P, graph=True, fused=form)
                xn = torch.randn(B, 768, device=dev) * 0.5
                for _ in range(3):
                    sess.step(xn)
                sess.pos.fill_(P)
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    sess.pos.fill_(P)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(20):
                        sess.graph.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / 20)
                row.append((sess.persistent, min(ts)))
                if sess.persistent:
                    assert int(sess.ws[1]) == 0, 'a poll timed out'
        print(f'batch {B} position {P}..{P + 20}: five launches per layer {row[0][1]:7.1f} us | persistent {row[1][1]:7.1f} us per step')
