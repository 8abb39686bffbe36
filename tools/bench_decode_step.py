#!/usr/bin/env python3
"""Where a cached ART-V decode step spends its time: each launch kind of the fused step alone (HIP events, 50 reps), the
whole 12-layer step eagerly and as a graph replay, and the complete per-token chain of DALLE._sample_cached."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mmvid_amd import _lib, ops

dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
m = bench.build_model(5, dev).eval()
tw = m.transformer


def timeit(fn, reps=50):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


E, F = 768, 3072
x = torch.randn(B, E, device=dev)
xf = torch.randn(B, F, device=dev)
sh = tw._sync_shadow()
blk = tw.transformer.resblocks[0]
ln = (blk.ln_1.weight, blk.ln_1.bias, 1e-5)
print(f'batch {B}')
print('gemv LN+qkv  768->2304: %.1f us' % timeit(lambda: ops.gemv_rows(x, sh[0], blk.attn.in_proj_bias, ln=ln, round_in=True)))
print('gemv out     768->768 : %.1f us' % timeit(lambda: ops.gemv_rows(x, sh[1], blk.attn.out_proj.bias, residual=x)))
print('gemv LN+fc   768->3072: %.1f us' % timeit(lambda: ops.gemv_rows(x, sh[2], blk.mlp.c_fc.bias, ln=ln, act=1, round_in=True)))
print('gemv proj   3072->768 : %.1f us' % timeit(lambda: ops.gemv_rows(xf, sh[3], blk.mlp.c_proj.bias, residual=x)))
cache = tw.new_kv_cache(B, 1152, dev)
cache.normal_()
for first in (129, 1100):
    for fused in (True, False):
        sess = tw.decode_session(cache, first, graph=False, fused=fused)
        sess.x.normal_()

        def step():
            sess.pos.fill_(first)
            sess._enqueue()

        t = timeit(step, 20)
        print(f'tower decode step at position {first}, fused={fused}: {t:.1f} us eager')
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                sess._enqueue()
        torch.cuda.current_stream().wait_stream(s)

        def rep():
            sess.pos.fill_(first)
            g.replay()

        print(f'   as a graph replay: {timeit(rep, 20):.1f} us')
text = torch.randint(1, 49408, (B, 64), device=dev)
vt = torch.randint(0, 1024, (B, 64), device=dev)
m.generate_images(text, visual=vt)
torch.cuda.synchronize()
t0 = time.perf_counter()
m.generate_images(text, visual=vt)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'generate_images: {dt * 1e3:.1f} ms per call = {dt / 1024 * 1e6:.1f} us per token step (incl. prefill + VQGAN decode of {16 * B} frames)')

# ---- platform floor: a dependent chain of 60 trivial kernels (one 64-thread block each), eager and as a graph replay
c = torch.zeros(1, device=dev)


def chain():
    for _ in range(60):
        ops.counter_add(c, 1.0)


print('60 dependent trivial kernels, eager: %.1f us' % timeit(chain, 20))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        chain()
torch.cuda.current_stream().wait_stream(s)
print('60 dependent trivial kernels, graph replay: %.1f us  (= the floor of a 60-launch decode step on this box)' % timeit(g.replay, 20))
