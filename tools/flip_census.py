#!/usr/bin/env python3
"""Token-index flip census of the VQGAN tokeniser's arithmetic modes (round 6).

Fresh full-size frames -- none of them a golden -- go through every mode of `VQGanVAE1024.get_codebook_indices` on the GPU:
default bf16, 'mixed', 'split', fp32 (`strict = True`); for the first `--oracle-frames` frames also through the CPU oracle
(`oracle/vqgan.py`, pinned bit for bit to the reference on tests/golden/vqgan_full16*.npz).  Per mode:

  * flips against the fp32 mode (all tokens) and against the oracle (the oracle's frames);
  * per token, gap / err: gap = the fp32 mode's top-2 distance gap, err = |gap(z_mode) - gap(z_fp32)| for the SAME two codes, both in
    fp64 from the fp32 z rows (only distance differences decide an argmin) -- the histogram says how far a mode is from flipping.

Two weight sets (seed 11 + synthetic codebook 0.5 N(0,1); seed 23 + the reference's initialisation U(-1/1024, 1/1024), quantize.py:254)
and two frame families (uniform noise, as the goldens; smooth fields = low-frequency noise upsampled, closer to video frames).

    python tools/flip_census.py [--frames 160] [--oracle-frames 16]          # 160 frames = 10,240 tokens per (weights, family)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

EDGES = [0, 1, 2, 4, 8, 64, 512, 4096, 1e30]


def frames_of(family, n, seed):
    g = torch.Generator().manual_seed(seed)
    if family == 'noise':
        return torch.rand(n, 3, 128, 128, generator=g)
    low = torch.rand(n, 3, 8, 8, generator=g)
    img = torch.nn.functional.interpolate(low, size=(128, 128), mode='bicubic', align_corners=False)
    img = img + 0.05 * torch.randn(n, 3, 128, 128, generator=g)
    return img.clamp_(0, 1)


def gap_table(z_ref, z_mode, e, idx_ref):
    """per token (gap of the reference z's best / runner-up codes, |error of that gap| under z_mode), fp64."""
    zr, zm = z_ref.double(), z_mode.double()
    D = (e * e).sum(1)[None, :] - 2.0 * zr @ e.t()
    rows = torch.arange(zr.shape[0])
    d1 = D[rows, idx_ref]
    D[rows, idx_ref] = float('inf')
    c2 = D.argmin(1)
    gap = D[rows, c2] - d1
    ev = e[c2] - e[idx_ref]  # gap(z) = |e2|^2 - |e1|^2 - 2 z.(e2 - e1): its error is 2 (z_mode - z_ref).(e2 - e1)
    err = (2.0 * ((zm - zr) * ev).sum(1)).abs()
    return gap, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=160)
    ap.add_argument('--oracle-frames', type=int, default=16)
    ap.add_argument('--chunk', type=int, default=32)
    a = ap.parse_args()
    from conftest import Golden, wide_vqgan_case
    from mmvid_amd.vae import VQGanVAE1024
    from oracle import vqgan as ov
    dev = torch.device('cuda', 0)
    modes = [('bf16', False), ('mixed', 'mixed'), ('split', 'split'), ('fp32', True)]
    print(f'# flip census: {a.frames} frames = {a.frames * 64} tokens per (weights, frame family); oracle on the first {a.oracle_frames}')
    print(f'# gap / err histogram edges: {EDGES[:-1]}')
    totals = {}
    for gname in ('vqgan_full16', 'vqgan_full16_refinit'):
        g = Golden(gname)
        sd, _ = wide_vqgan_case(g)
        vae = VQGanVAE1024(None, 128)
        vae.load_state_dict(sd)
        vae = vae.to(dev)
        vae.image_size = 128
        e = sd['model.quantize.embedding.weight'].double()
        for fam in ('noise', 'smooth'):
            img = frames_of(fam, a.frames, 1000 + g.meta['seed'])
            t0 = time.time()
            with torch.no_grad():
                oi = ov.get_codebook_indices(sd, img[:a.oracle_frames], 128).reshape(-1)
            t_or = time.time() - t0
            res = {}
            for name, strict in modes:
                vae.strict = strict
                zs, ids = [], []
                for i in range(0, a.frames, a.chunk):
                    x = img[i:i + a.chunk].to(dev)
                    ids.append(vae.get_codebook_indices(x).reshape(-1).cpu())
                    zs.append(vae.encode_z(x).reshape(-1, e.shape[1]).cpu())
                res[name] = (torch.cat(ids), torch.cat(zs))
            vae.strict = False
            i32, z32 = res['fp32']
            no = oi.numel()
            print(f'== weights {gname} (seed {g.meta["seed"]}, codebook {g.meta["codebook"]}), frames {fam}: {i32.numel()} tokens, '
                  f'{i32.unique().numel()} distinct codes; oracle {no} tokens in {t_or:.1f} s')
            print(f'   fp32 mode vs oracle: {int((i32[:no] != oi).sum())} flips of {no}')
            for name, _ in modes[:-1]:
                im, zm = res[name]
                gap, err = gap_table(z32, zm, e, i32)
                r = (gap / err.clamp_min(1e-30)).numpy()
                hist = np.histogram(r, EDGES)[0].tolist()
                fl = im != i32
                flo = im[:no] != oi
                key = (name, g.meta['codebook'])
                t = totals.setdefault(key, [0, 0, 0, 0, 0])
                t[0] += int(fl.sum()); t[1] += fl.numel(); t[2] += int(flo.sum()); t[3] += no; t[4] += int((r < 8).sum())
                worst = r[fl.numpy()].max() if fl.any() else float('nan')
                print(f'   {name:6s}: {int(fl.sum()):4d} flips vs fp32 ({100 * fl.float().mean().item():.3f} %), {int(flo.sum()):3d} of {no} vs oracle; '
                      f'max |dz| {(zm - z32).abs().max().item():.2e}; gap/err min {r.min():.2f} median {np.median(r):.0f}; '
                      f'histogram {hist}; largest gap/err among flips {worst:.2f}')
    print('# totals over both frame families')
    for (name, cbk), t in totals.items():
        print(f'#   {name:6s} codebook {cbk:14s}: {t[0]} flips of {t[1]} vs fp32 ({100 * t[0] / t[1]:.3f} %), {t[2]} of {t[3]} vs oracle, '
              f'{t[4]} tokens with gap/err < 8')


if __name__ == '__main__':
    main()
