#!/usr/bin/env python3
"""Kernel timings at the training step's shapes, HIP-event timed: attention forward / backward, LayerNorm forward / backward,
GroupNorm + swish.  (Round 5 used this file for the A/Bs of the packed softmax arithmetic, the tail splits, the pipelined LayerNorm
backward and the fused GroupNorm apply: profiles/r05_micro_attention_pk_tail_ln_gn.log.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev, bf = 'cuda', torch.bfloat16
st = ops._stream


def attention():
    B, L, H, E = 18, 579, 12, 768
    torch.manual_seed(0)
    qkv = (torch.randn(B * L, 3 * E, device=dev) * 0.5).to(bf)
    dO = (torch.randn(B * L, E, device=dev) * 0.1).to(bf)
    out = torch.empty(B * L, E, device=dev, dtype=bf)
    lse, delta = torch.empty(B * H * L, device=dev), torch.empty(B * H * L, device=dev)
    dqkv = torch.empty(B * L, 3 * E, device=dev, dtype=bf)
    rows = (2, 65, 65, 66, 66)

    def fwd():
        _lib.call('mmvid_attention_fwd', ops._p(qkv), 3 * E, B, L, H, E, 0.125, *rows, ops._p(out), E, ops._p(lse), st())

    def bwd():
        _lib.call('mmvid_attention_bwd_bias', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E, 0.125, *rows,
                  ops._p(dqkv), 3 * E, None, st())

    fl = 4.0 * B * H * L * L * 64
    fwd()
    tf, tb = timeit(fwd, 40), timeit(bwd, 40)
    print(f'attention B={B} L={L} H={H}: fwd {tf*1e3:6.1f} us {fl/tf/1e9:6.1f} TF | bwd (dQ + dK/dV) {tb*1e3:6.1f} us {2.5*fl/tb/1e9:6.1f} TF')


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

    def run():
        _lib.call('mmvid_layernorm_bwd_partial', ops._p(dy16), 1, E, ops._p(x), E, ops._p(mean), ops._p(rstd), ops._p(w), M, E, ops._p(g), E, 1,
                  ops._p(gb), 1, 1, 1, ops._p(ws), ws.numel(), ctypes.byref(nb), st())

    t = timeit(run, 50)
    alg = M * E * (2 + 4 + 4 + 4 + 2)
    print(f'LN bwd (the tower\'s call: bf16 dy, += g, bf16 copy, partial rows): {t*1e3:6.1f} us  {alg/t/1e9:6.2f} TB/s algorithmic')


def groupnorm():
    for N, H, C, dt in ((54, 128, 128, bf), (54, 128, 128, torch.float32), (54, 64, 128, bf), (54, 32, 256, bf)):
        x = torch.randn(N, H, H, C, device=dev).to(dt)
        w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        stats = ops.gn_stats_buffer(N, H * H, C, dev)
        stats.normal_()  # (timing only: the partial sums are whatever is there)
        t = timeit(lambda: ops.groupnorm_swish(x, w, b, stats=stats, stats_block=128), 30)
        by = N * H * H * C * (x.element_size() + 2)
        print(f'GroupNorm+swish {N}x{H}x{H}x{C} {"bf16" if dt == bf else "f32 "} in, conv-fused statistics: {t*1e3:6.1f} us {by/t/1e9:5.2f} TB/s')


if __name__ == '__main__':
    which = sys.argv[1:] or ['attention', 'layernorm', 'groupnorm']
    for wname in which:
        globals()[wname]()
