#!/usr/bin/env python3
"""Per-op timing of one planned VQGAN encode (the step's 96 frames): every op of the plan is launched on its own between
two HIP events (after one full warm-up run), then grouped by (op kind, geometry).  Prints ms, TFLOP/s and the
algorithmic HBM GB/s of every group -- where the 6+ ms of the encoder go, layer by layer."""
import ctypes
import os
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd import _lib, ops
from mmvid_amd.vae import VQGanVAE1024

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device('cuda', 0)
torch.manual_seed(0)
vae = VQGanVAE1024(None, 128).to(dev)
vae.image_size = 128
MODE = sys.argv[2] if len(sys.argv) > 2 else 'bf16'  # bf16 | strict | split | mixed
vae.strict = {'bf16': False, 'strict': True}.get(MODE, MODE)
img = torch.rand(N, 3, 128, 128, device=dev)
vae.get_codebook_indices(img)
plan = vae._plan('enc', N, 128)
idx = torch.empty(N, 64, device=dev, dtype=torch.int64)
for i, field, name in plan.patches:
    setattr(plan.ops[i], field, (img if name == 'img' else idx).data_ptr())
KIND = ['img', 'conv', 'gn', 'cast', 'attn', 'vq', 'gather', 'nchw', 'ext']
rows = OrderedDict()
reps = 3
for i in range(len(plan.ops)):
    o = plan.ops[i]
    sub = ctypes.cast(ctypes.byref(plan.ops, i * ctypes.sizeof(_lib.VqganOp)), ctypes.POINTER(_lib.VqganOp))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.call('mmvid_vqgan_run', sub, 1, ops._p(plan.arena), ops._stream())
    a.record()
    for _ in range(reps):
        _lib.call('mmvid_vqgan_run', sub, 1, ops._p(plan.arena), ops._stream())
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    kind = KIND[o.op]
    H, W = o.H, o.W
    Ho, Wo = (H // 2, W // 2) if (kind == 'conv' and o.mode == 1) else ((2 * H, 2 * W) if (kind == 'conv' and o.mode == 2) else (H, W))
    flops = byts = 0.0
    if kind == 'conv':
        taps = 1 if o.mode == 3 else 9
        flops = 2.0 * N * Ho * Wo * o.Cout * taps * o.C
        in_b = 4 if (o.flags & 64 and not o.flags & 128) else 2  # pair planes / one fp16 or bf16 plane
        byts = N * H * W * o.C * in_b + N * Ho * Wo * o.Cout * ((2 if o.out_bf16 >= 0 else 0) + (4 if o.out_f32 >= 0 else 0)) + \
            (N * Ho * Wo * o.Cout * (4 if (o.flags & 1 or o.flags & 64) else 2) if o.in1 >= 0 else 0)
        key = (kind, f'm{o.mode} {H}x{W} {o.C}->{o.Cout}' + (' +res' if o.in1 >= 0 else '') + (' f32out' if o.out_f32 >= 0 else '') +
               (' +bf16' if o.out_f32 >= 0 and o.out_bf16 >= 0 else '') + (' +gnstats' if o.flags & 4 else '') + (' f16' if o.flags & 128 else (' pair' if o.flags & 64 else '')) +
               (' splitk' if o.flags & 32 else ''))
    elif kind == 'gn':
        f32in = bool(o.flags & 1 or o.flags & 64)
        outb = 4 if (o.flags & 64 and not o.flags & 128) else 2
        byts = N * H * W * o.C * ((4 if f32in else 2) + outb)
        key = (kind, f'{H}x{W} C{o.C} ' + ('f32in' if f32in else 'bf16in') + (' fused-stats' if o.flags & 2 else ' own-stats') +
               (' ->f16' if o.flags & 128 else (' ->pair' if o.flags & 64 else '')))
    elif kind == 'cast':
        byts = N * H * W * o.C * 8
        key = (kind, f'{H}x{W} C{o.C}')
    else:
        key = (kind, f'{H}x{W} C{o.C}')
    r = rows.setdefault(key, [0, 0.0, 0.0, 0.0])
    r[0] += 1
    r[1] += ms
    r[2] += flops
    r[3] += byts
tot = sum(r[1] for r in rows.values())
print(f'VQGAN encode of {N} frames, mode {MODE}, per-op timing (sum {tot:.3f} ms; ops run back to back are faster than this sum by the launch gaps)')
print(f'{"op":6s} {"geometry":52s} {"n":>3s} {"ms":>8s} {"%":>5s} {"TFLOP/s":>8s} {"GB/s":>7s}')
for (kind, geo), (n, ms, fl, by) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f'{kind:6s} {geo:52s} {n:3d} {ms:8.3f} {100 * ms / tot:5.1f} {fl / ms / 1e9 if fl else 0:8.1f} {by / ms / 1e6 if by else 0:7.0f}')
for kind in ('conv', 'gn', 'attn', 'cast'):
    ms = sum(r[1] for (k, _), r in rows.items() if k == kind)
    fl = sum(r[2] for (k, _), r in rows.items() if k == kind)
    print(f'total {kind}: {ms:.3f} ms' + (f', {fl / ms / 1e9:.1f} TFLOP/s' if fl else ''))
