#!/usr/bin/env python3
"""A/B of the 256x128 GEMM epilogues (option gemm_epi: 0 = LDS-staged, 1 = register-direct + early prologue) on the tower's
shapes: time, TFLOP/s and bit-equality of every output.  python tools/bench_gemm_epi.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from bench_gemm import timeit

dev, bf = 'cuda', torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10422
torch.manual_seed(0)


def cases():
    E = 768
    X = torch.randn(M, E, device=dev).to(bf)
    H = torch.randn(M, 4 * E, device=dev).to(bf)
    res = torch.randn(M, E, device=dev)
    for name, A, N, K, kw in (
            ('qkv fwd   (bias, bf16)', X, 3 * E, E, dict(bias=True)),
            ('out fwd   (bias, +res, f32)', X, E, E, dict(bias=True, residual=res, out_dtype=torch.float32)),
            ('fc fwd    (bias, gelu, pre saved)', X, 4 * E, E, dict(bias=True, act=1, save=True)),
            ('proj fwd  (bias, +res, f32)', H, E, 4 * E, dict(bias=True, residual=res, out_dtype=torch.float32)),
            ('dX qkv    (NN, f32)', torch.randn(M, 3 * E, device=dev).to(bf), E, 3 * E, dict(km=True, out_dtype=torch.float32)),
            ('dX proj   (NN, dact, colsum, bf16)', X, 4 * E, E, dict(km=True, dact=True, colsum=True)),
            ('dX fc     (NN, acc f32)', H, E, 4 * E, dict(km=True, out_dtype=torch.float32, accumulate=True)),
    ):
        yield name, A, N, K, kw


for name, A, N, K, kw in cases():
    W = (torch.randn(K, N, device=dev) * 0.03).to(bf) if kw.get('km') else (torch.randn(N, K, device=dev) * 0.03).to(bf)
    bias = torch.randn(N, device=dev) * 0.1 if kw.get('bias') else None
    pre_in = torch.randn(M, N, device=dev).to(bf) if kw.get('dact') else None
    outs = {}
    row = f'{name:34s} {M}x{N}x{K}:'
    for epi in (0, 1, 4, 3):
        _lib.call('mmvid_set_option', b'gemm_epi', min(epi, 1))
        _lib.call('mmvid_set_option', b'gemm_loader', 1 if epi >= 3 else 0)
        _lib.call('mmvid_set_option', b'gemm_groupn', 0 if epi == 4 else 1)
        save = torch.zeros(M, N, device=dev, dtype=bf) if kw.get('save') else None
        base = torch.randn(M, N, device=dev, generator=torch.Generator(dev).manual_seed(3)) if kw.get('accumulate') else None

        cs = torch.zeros(N, device=dev) if kw.get('colsum') else None

        def run(o=None):
            return ops.gemm(A, W, b_kmajor=bool(kw.get('km')), bias=bias, residual=kw.get('residual'), dact_pre=pre_in, save_pre=save,
                            act=kw.get('act', 0), out_dtype=kw.get('out_dtype', bf), out=o, accumulate=bool(kw.get('accumulate')), colsum=cs)

        y = run(base.clone() if base is not None else None)
        outs[epi] = (y.clone(), save.clone() if save is not None else None)
        if cs is not None:
            ref_cs = y.float().sum(0)  # (of the bf16-rounded result: ~1e-3 relative to the fp32 sums the kernel takes)
            err = ((cs - ref_cs).abs().max() / ref_cs.abs().max()).item()
            assert err < 5e-3, (name, epi, err)
            row += f' [colsum err {err:.1e}]'
        scratch = base.clone() if base is not None else None
        t = timeit(lambda: run(scratch))
        fl = 2.0 * M * N * K
        row += f'  {["lds", "direct", "defer", "LOADER+groups", "LOADER"][epi]}: {t * 1e3:6.1f} us {fl / t / 1e9:6.1f} TF'
    same = all(torch.equal(outs[0][0], outs[e][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[e][1])) for e in (1, 4, 3))
    fin = bool(torch.isfinite(outs[1][0].float()).all())
    row += f'  | bit-identical {same} finite {fin}'
    print(row, flush=True)
    assert same and fin, name
_lib.call('mmvid_set_option', b'gemm_epi', 1)
# ragged edges: M, N not multiples of the tile; bias; small K (nt = 1, 2)
for (m, n, k) in ((1000, 136, 64), (777, 2304, 128), (2561, 776, 200), (300, 3072, 768), (70000, 256, 64), (9000, 1280, 192)):
    A = torch.randn(m, k, device=dev).to(bf)
    Wt = (torch.randn(n, k, device=dev) * 0.05).to(bf)
    b = torch.randn(n, device=dev)
    ys = []
    for epi in (0, 1, 3):
        _lib.call('mmvid_set_option', b'gemm_epi', min(epi, 1))
        _lib.call('mmvid_set_option', b'gemm_loader', 1 if epi == 3 else 0)
        _lib.call('mmvid_set_option', b'gemm_tile', 256)
        ys.append(ops.gemm(A, Wt, bias=b, out_dtype=torch.float32))
    ref = A.float() @ Wt.float().t() + b
    err = (ys[1] - ref).abs().max().item() / ref.abs().max().item()
    yb = []
    for epi in (0, 2, 3):  # bf16 output: the deferred path when there are more tiles than CUs; the loader-wave kernel
        _lib.call('mmvid_set_option', b'gemm_epi', min(epi, 2) if epi != 3 else 1)
        _lib.call('mmvid_set_option', b'gemm_loader', 1 if epi == 3 else 0)
        yb.append(ops.gemm(A, Wt, bias=b))
    print(f'ragged {m}x{n}x{k}: bit-identical {torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])} (bf16 deferred {torch.equal(yb[0], yb[1])} loader {torch.equal(yb[0], yb[2])}), rel err vs fp32 torch {err:.2e}')
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]) and torch.equal(yb[0], yb[1]) and torch.equal(yb[0], yb[2]) and err < 1e-2
_lib.call('mmvid_set_option', b'gemm_tile', 0)
_lib.call('mmvid_set_option', b'gemm_epi', 1)
_lib.call('mmvid_set_option', b'gemm_loader', 1)
print('ok')
# dW GEMMs (k-major operands, split-K slabs + fixed-order reduce): LDS-staged block form vs the loader-wave form
for name, N, K in (('dW in_proj', 2304, 768), ('dW out', 768, 768), ('dW fc', 3072, 768), ('dW proj', 768, 3072)):
    dY = torch.randn(M, N, device=dev).to(bf)
    Xk = torch.randn(M, K, device=dev).to(bf)
    res = []
    row = f'{name:34s} {N}x{K}x{M}:'
    for loader in (0, 1):
        _lib.call('mmvid_set_option', b'gemm_loader', loader)
        dW = torch.zeros(N, K, device=dev)
        ops.gemm_dw(dY, Xk, dW)
        res.append(dW.clone())
        t = timeit(lambda: ops.gemm_dw(dY, Xk, dW))
        row += f'  {"LOADER" if loader else "lds"}: {t * 1e3:6.1f} us {2.0 * M * N * K / t / 1e9:6.1f} TF'
    ref = dY.float().t() @ Xk.float()
    err = ((res[1] - ref).abs().max() / ref.abs().max()).item()
    print(row + f'  | bit-identical {torch.equal(res[0], res[1])} rel err vs fp32 torch {err:.1e}', flush=True)
    assert torch.equal(res[0], res[1]) and err < 1e-2
_lib.call('mmvid_set_option', b'gemm_loader', 1)
print('dW ok')
