#!/usr/bin/env python3
"""Round-5 kernel A/Bs at the training step's shapes, HIP-event timed: attention forward / backward (workspace = tail splits on or
off, packed or single-lane softmax arithmetic), LayerNorm backward (pipelined or generic kernel), GroupNorm (finalisation inside the
apply launch or separate), LayerNorm forward."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev, bf = 'cuda', torch.bfloat16
st = ops._stream


def opt(name, v):
    _lib.call('mmvid_set_option', name.encode(), v)


def attention():
    B, L, H, E = 18, 579, 12, 768
    torch.manual_seed(0)
    qkv = (torch.randn(B * L, 3 * E, device=dev) * 0.5).to(bf)
    dO = (torch.randn(B * L, E, device=dev) * 0.1).to(bf)
    out = torch.empty(B * L, E, device=dev, dtype=bf)
    lse, delta = torch.empty(B * H * L, device=dev), torch.empty(B * H * L, device=dev)
    dqkv = torch.empty(B * L, 3 * E, device=dev, dtype=bf)
    nws = _lib.load().mmvid_attention_bwd_workspace_bytes(B, L, H)
    ws = torch.empty(nws, device=dev, dtype=torch.uint8)
    rows = (2, 65, 65, 66, 66)

    def fwd(w):
        _lib.call('mmvid_attention_fwd', ops._p(qkv), 3 * E, B, L, H, E, 0.125, *rows, ops._p(out), E, ops._p(lse), st())

    def bwd(w):
        _lib.call('mmvid_attention_bwd_ws', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E, 0.125, *rows,
                  ops._p(dqkv), 3 * E, None, ops._p(ws) if w else None, nws if w else 0, st())

    fl = 4.0 * B * H * L * L * 64
    fwd(1)
    for pk in (3, 0):  # bit 0: forward packed, bit 1: dQ packed
        opt('attn_pk', pk)
        for tail in (0, 2, 4, 6):  # bit 1: dQ split, bit 2: dK / dV split
            opt('attn_tail', tail)
            tf, tb = timeit(lambda: fwd(1), 40), timeit(lambda: bwd(1), 40)
            print(f'attention pk={pk} tail bits={tail}: fwd {tf*1e3:6.1f} us {fl/tf/1e9:6.1f} TF | bwd (dQ + dK/dV) {tb*1e3:6.1f} us {2.5*fl/tb/1e9:6.1f} TF')
    opt('attn_pk', 3), opt('attn_tail', 7)


def layernorm():
    M, E = 10422, 768
    torch.manual_seed(0)
    x = torch.randn(M, E, device=dev)
    w, b = torch.randn(E, device=dev), torch.randn(E, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, w, b), 50)
    print(f'LN fwd: {t*1e3:6.1f} us  {(M*E*6 + M*8)/t/1e9:6.2f} TB/s')
    y, mean, rstd = ops.layernorm_fwd(x, w, b)
    dy16 = torch.randn(M, E, device=dev).bfloat16()
    g = torch.randn(M, E, device=dev)
    gb = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(512 * 3 * E, device=dev)
    import ctypes
    nb = ctypes.c_int()
    for fast in (0, 1):
        opt('ln_fast', fast)

        def run():
            _lib.call('mmvid_layernorm_bwd_partial', ops._p(dy16), 1, E, ops._p(x), E, ops._p(mean), ops._p(rstd), ops._p(w), M, E, ops._p(g), E, 1,
                      ops._p(gb), 1, 1, 1, ops._p(ws), ws.numel(), ctypes.byref(nb), st())

        t = timeit(run, 50)
        alg = M * E * (2 + 4 + 4 + 4 + 2)
        print(f'LN bwd (the tower\'s call: bf16 dy, += g, bf16 copy, partial rows) ln_fast={fast}: {t*1e3:6.1f} us  {alg/t/1e9:6.2f} TB/s algorithmic')
    opt('ln_fast', 1)


def groupnorm():
    for N, H, C, dt in ((54, 128, 128, bf), (54, 128, 128, torch.float32), (54, 64, 128, bf), (54, 32, 256, bf)):
        x = torch.randn(N, H, H, C, device=dev).to(dt)
        w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        stats = ops.gn_stats_buffer(N, H * H, C, dev)
        stats.normal_()  # (timing only: the partial sums are whatever is there)
        for fused in (0, 1):
            opt('gn_fused', fused)
            t = timeit(lambda: ops.groupnorm_swish(x, w, b, stats=stats, stats_block=128), 30)
            by = N * H * H * C * (x.element_size() + 2)
            print(f'GroupNorm+swish {N}x{H}x{H}x{C} {"bf16" if dt == bf else "f32 "} in, conv-fused statistics, gn_fused={fused}: {t*1e3:6.1f} us {by/t/1e9:5.2f} TB/s')
    opt('gn_fused', 1)


if __name__ == '__main__':
    which = sys.argv[1:] or ['attention', 'layernorm', 'groupnorm']
    for wname in which:
        globals()[wname]()
