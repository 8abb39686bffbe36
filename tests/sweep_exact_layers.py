#!/usr/bin/env python3
"""Per-layer precision sensitivity of the VQGAN encoder's token indices (round-4 review, item 4) -- CPU only, not a test.

The exact-index mode (`vae.strict = 'split'`) runs EVERY convolution as three bf16 products of hi/lo operand pairs.  This script asks
where that is needed: on the full-size encoder with the golden synthetic weights it runs the oracle's functional encoder
(oracle/vqgan.py) in fp64 with the operands of chosen convolutions rounded the way a cheaper form would round them --

    'split' : x = hi + lo (two bf16), w = hi + lo, products hi.hi + lo.hi + hi.lo   (what csrc/conv*.hip computes with terms = 3)
    'bf16'  : x, w rounded to bf16, one product
    'fp16'  : x, w rounded to fp16, one product (the matrix pipe's f16 rate equals its bf16 rate)
    'f64'   : nothing rounded

-- everything else (accumulation, GroupNorm, attention) stays fp64, so the numbers are the error of the operand formats alone.  Per
configuration: max |dz|, and the acceptance quantity of tests/test_round3_gpu.py::test_split_index_safety_margin -- over all tokens the
minimum of (the reference's top-2 distance gap) / (the error of that gap), which must stay above 8.

usage: python tests/sweep_exact_layers.py [extra frames]   (the golden's frames + 4 fresh ones by default; about a minute on 8 cores)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import vqgan as ov
from oracle.synth import synth_input, synth_tensor

MODE = {}      # layer name -> 'f64' | 'split' | 'bf16' | 'fp16'
DEFAULT = ['split']
SEEN = []      # (name, flops) in call order


def _round(t, fmt):
    if fmt == 'bf16':
        return t.float().bfloat16().double()
    if fmt == 'fp16':
        return t.float().half().double()
    raise ValueError(fmt)


def _conv(sd, p, x, stride=1, padding=1):
    w, b = sd[p + '.weight'], sd[p + '.bias']
    mode = MODE.get(p, DEFAULT[0])
    if not SEEN or all(n != p for n, _ in SEEN):
        ho = (x.shape[2] + 2 * padding - w.shape[2]) // stride + 1
        SEEN.append((p, 2.0 * ho * ho * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]))
    if mode == 'f64':
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    if mode == 'split':  # operands as fp32 -> (hi, lo) bf16 pairs; the lo.lo product is dropped
        x32, w32 = x.float(), w.float()
        xh, wh = x32.bfloat16().float(), w32.bfloat16().float()
        xl, wl = (x32 - xh).bfloat16().double(), (w32 - wh).bfloat16().double()
        xh, wh = xh.double(), wh.double()
        return F.conv2d(xh + xl, wh, b, stride=stride, padding=padding) + F.conv2d(xh, wl, None, stride=stride, padding=padding)
    return F.conv2d(_round(x, mode), _round(w, mode), b, stride=stride, padding=padding)


ov._conv = _conv


def main():
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(min(8, os.cpu_count() or 1))  # (more threads than cores with fp64 convolutions: minutes of spinning)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vqgan_full.npz'))
    manifest = json.loads(bytes(g['manifest']).decode())  # the reference state_dict's key -> shape list
    sd = {k: synth_tensor(k, tuple(s), 11).double() for k, s in manifest}
    meta = json.loads(bytes(g['meta']).decode())
    ng = int(meta['n'])  # the golden's own frames first (the ones test_split_index_safety_margin looks at), then fresh ones
    img = torch.cat([synth_input('img', (ng, 3, 128, 128), 11, 'uniform'), synth_input('img_sweep', (nframes, 3, 128, 128), 11, 'uniform')]).double()
    gold = torch.arange(img.shape[0] * 64) < ng * 64
    e = sd['model.quantize.embedding.weight']

    def run():
        z = ov.encode_z(sd, img, 128)
        return z.permute(0, 2, 3, 1).reshape(-1, z.shape[1])

    def dist(z):
        return (e * e).sum(1)[None, :] - 2.0 * z @ e.t()

    DEFAULT[0] = 'f64'
    t0 = time.time()
    zr = run()
    print(f'# fp64 encoder pass: {time.time() - t0:.1f} s, {zr.shape[0]} tokens, |z| max {zr.abs().max():.3f} rms {zr.pow(2).mean().sqrt():.3f}')
    Dr = dist(zr)
    rows = torch.arange(zr.shape[0])
    idx = Dr.argmin(1)
    d1 = Dr[rows, idx]
    Dm = Dr.clone()
    Dm[rows, idx] = float('inf')
    c2 = Dm.argmin(1)
    gap_r = Dr[rows, c2] - d1
    print(f'# reference top-2 gap: min {gap_r.min():.3e} median {gap_r.median():.3e}')
    layers = list(SEEN)
    total = sum(f for _, f in layers)

    def measure(label, modes, default):
        MODE.clear(), MODE.update(modes)
        DEFAULT[0] = default
        zs = run()
        Ds = dist(zs)
        err = ((Ds[rows, c2] - Ds[rows, idx]) - gap_r).abs().clamp_min(1e-30)
        r = gap_r / err
        flips = int((Ds.argmin(1) != idx).sum())
        # matrix-pipe work relative to plain bf16 everywhere: split = 3 products
        cost = sum(f * (3 if modes.get(n, default) == 'split' else 1) for n, f in layers) / total
        print(f'{label:58s} max|dz| {float((zs - zr).abs().max()):.3e}  min gap/err: golden frames {float(r[gold].min()):8.1f}, all {float(r.min()):8.1f}; '
              f'tokens below 8: {int((r < 8).sum()):3d} of {r.numel()}, flips {flips:2d}; MFMA work {cost:.2f}x')
        sys.stdout.flush()
        return float(r.min())

    print('# layer list (call order): ' + ', '.join(f'{n.replace("encoder.", "")} {100 * f / total:.1f}%' for n, f in layers))
    measure('every convolution split (the shipped exact mode)', {}, 'split')
    measure('every convolution plain bf16 (default mode, operands only)', {}, 'bf16')
    measure('every convolution plain fp16', {}, 'fp16')
    for fmt in ('bf16', 'fp16'):
        for n, f in layers:
            if f / total < 0.02 and fmt == 'bf16':
                continue
            measure(f'one layer {fmt}: {n.replace("encoder.", "")} ({100 * f / total:.1f}% of the work)', {n: fmt}, 'split')
    def blocks(*levels):  # the 3x3 convolutions of the residual blocks of these levels (what vae.strict = 'mixed' switches to fp16)
        return [n for n, _ in layers if any(f'.down.{k}.block' in n for k in levels) and 'nin_shortcut' not in n]
    measure("levels 0-1 residual-block 3x3 convolutions fp16, rest split", {n: 'fp16' for n in blocks(0, 1)}, 'split')
    measure("levels 0-2 residual-block 3x3 convolutions fp16, rest split  [= vae.strict = 'mixed']", {n: 'fp16' for n in blocks(0, 1, 2)}, 'split')
    lvl0 = [n for n, _ in layers if '.down.0.block' in n]
    lvl1 = [n for n, _ in layers if '.down.1.block' in n]
    ds0 = [n for n, _ in layers if 'down.0.downsample' in n]
    for fmt in ('bf16', 'fp16'):
        measure(f'level 0 (128x128) residual blocks {fmt}, rest split', {n: fmt for n in lvl0}, 'split')
        measure(f'levels 0-1 residual blocks + downsample {fmt}, rest split', {n: fmt for n in lvl0 + lvl1 + ds0}, 'split')


if __name__ == '__main__':
    main()
