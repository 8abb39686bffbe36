#!/usr/bin/env python3
"""Can the 56-block tail of the attention forward be hidden by launching it as a SECOND kernel on another stream?  Times 17 sequences
on one stream + 1 sequence on a second stream (same buffers, disjoint rows) against the single 18-sequence launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops

dev, bf = 'cuda', torch.bfloat16
L, H, E = 579, 12, 768
torch.manual_seed(0)
B = 18
qkv = (torch.randn(B * L, 3 * E, device=dev) * 0.5).to(bf)
out = torch.empty(B * L, E, device=dev, dtype=bf)
lse = torch.empty(B * H * L, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def launch(b0, nb, stream):
    with torch.cuda.stream(stream):
        q = qkv[b0 * L:]
        o = out[b0 * L:]
        l = lse[b0 * H * L:]
        _lib.call('mmvid_attention_fwd', ops._p(q), 3 * E, nb, L, H, E, 0.125, 2, 65, 65, 66, 66, ops._p(o), E, ops._p(l), ops._stream())


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
        a.record()
    for _ in range(reps):
        fn()
    with torch.cuda.stream(s1):
        s1.wait_stream(s2)
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def single():
    launch(0, 18, s1)


def split(n2):
    def f():
        s2.wait_stream(s1)          # (the pair starts together: the previous pair is complete on both streams)
        launch(0, 18 - n2, s1)
        launch(18 - n2, n2, s2)
        s1.wait_stream(s2)
    return f


print(f'one launch of 18 sequences (1,080 blocks): {timed(single):.1f} us')
print(f'one launch of 17 sequences (1,020 blocks): {timed(lambda: launch(0, 17, s1)):.1f} us')
for n2 in (1, 2, 4):
    print(f'{18 - n2} sequences on stream 1 + {n2} on stream 2, joined: {timed(split(n2)):.1f} us')
