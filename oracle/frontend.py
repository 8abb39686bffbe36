"""CPU restatement of the stochastic front-end of BERT.forward (TEST INFRASTRUCTURE).

The reference draws these on the host inside forward(), from numpy / `random` / torch / torchvision generators:
  msm_masks        mmvid_pytorch/dalle_bert.py:992-1029
  RandomErasing    torchvision.transforms.RandomErasing as constructed at dalle_bert.py:290-294 / 427-432 and
                   dalle_artv.py:229-232 -- THIRD PARTY (torchvision is absent from /root/reference and unpinned in its
                   requirements.txt): restated from the published semantics, "parity unpinned" for the box distribution
  warp             dalle_bert.py:93-238 (frame from another sample / frame shuffle / colour shift / affine warp)
PINNED (tests/test_oracle_golden.py::test_frontend_*): given the decisions the reference's generators drew (recovered by
tools/make_golden.py::case_frontend by replaying the draws), `apply_warp`, `affine_warp`, `color_shift`,
`video_color_shift`, `swap_halves` and `build_msm_mask` reproduce the outputs of the reference's own warp / warp_with_affine /
warp_with_color / warp_video_with_color / swap / MSM loop (tests/golden/frontend.npz).  What stays unpinned is only the
distribution of torchvision's RandomErasing box (third party).
The product draws the same DISTRIBUTIONS on the device (mmvid_amd/csrc/frontend.hip) from a counter-based generator;
bit-level agreement of two different generators is impossible, so tests compare statistics (strategy frequencies, box
area / aspect / position moments, keep rates) and compare the deterministic parts (the affine resampling for given
parameters) value by value.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def erasing_box(rng, H, W, scale, ratio):
    """torchvision RandomErasing.get_params: (i, j, h, w) or None when 10 attempts found no fitting box."""
    area = H * W
    l0, l1 = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        ea = area * rng.uniform(scale[0], scale[1])
        ar = math.exp(rng.uniform(l0, l1))
        h, w = int(round(math.sqrt(ea * ar))), int(round(math.sqrt(ea / ar)))
        if not (h < H and w < W):
            continue
        return int(rng.randint(0, H - h + 1)), int(rng.randint(0, W - w + 1)), h, w
    return None


def build_msm_mask(strategy, T, f, bern=None, box=None, keep_frames=()):
    """One sample's mask1 (True = visible) from the decisions of dalle_bert.py:996-1026: strategy 1 keeps where the Bernoulli
    draw `bern` is 1, 2 hides everything, 3 hides the RandomErasing box (i, j, h, w), 4 shows only the box; afterwards the
    frames in keep_frames are made fully visible (1022-1026).  -> (mask bool [T*f*f], not_fully_masked)."""
    n = f * f
    if strategy == 1:
        m = np.asarray(bern).reshape(-1) == 1
    elif strategy == 2:
        m = np.zeros(T * n, bool)
    else:
        one = np.ones((T, f, f), bool)
        if box is not None:
            i0, j0, h, w = box
            one[:, i0:i0 + h, j0:j0 + w] = False
        m = one.reshape(-1) if strategy == 3 else ~one.reshape(-1)
    m = m.copy()
    for tt in keep_frames:
        m[n * tt:n * (tt + 1)] = True
    return m, 0.0 if strategy == 2 else 1.0


def color_shift(frame, shift, num):
    """warp_with_color (dalle_bert.py:124-135) for a drawn (c_shift, num): frame [C,H,W]; num 0 = every channel."""
    m = torch.zeros_like(frame)
    if num == 0:
        m += shift
    else:
        m[num - 1] += shift
    return torch.clamp(frame + m, 0, 1)


def video_color_shift(video, params):
    """warp_video_with_color (dalle_bert.py:140-158): video [n,t,C,H,W], one (c_shift, num) per sample for all its frames."""
    out = []
    for x, (shift, num) in zip(video, params):
        m = torch.zeros_like(x)
        if int(num) == 0:
            m += float(shift)
        else:
            m[:, int(num) - 1] += float(shift)
        out.append(torch.clamp(x + m, 0, 1))
    return torch.stack(out)


def swap_halves(x):
    """swap(tensor, 0) for an even batch (dalle_bert.py:110-113): the two halves exchanged."""
    h = x.shape[0] // 2
    return torch.cat((x[h:], x[:h]), 0)


def msm_masks(rng, B, T, f, strategy_prob, bernoulli_prob, pc_prob=0.0):
    """dalle_bert.py:992-1029 -> (mask1 bool [B, T*f*f] (True = visible), not_fully_masked [B], strategies [B])."""
    n = f * f
    masks, nfm, strat = [], np.ones(B, np.float32), []
    for i in range(B):
        which = int(rng.choice([1, 2, 3, 4], p=strategy_prob))
        strat.append(which)
        if which == 1:
            p = rng.uniform(*bernoulli_prob)
            m = rng.random_sample(T * n) < p
        elif which == 2:
            nfm[i] = 0
            m = np.zeros(T * n, bool)
        else:
            one = np.ones((T, f, f), bool)
            box = erasing_box(rng, f, f, (0.2, 0.8), (0.5, 2.0))
            if box is not None:
                i0, j0, h, w = box
                one[:, i0:i0 + h, j0:j0 + w] = False
            m = one.reshape(-1) if which == 3 else ~one.reshape(-1)
        if pc_prob > 0 and rng.random_sample() < pc_prob:
            t_overlap = int(rng.randint(1, T // 2 + 1))
            for tt in rng.choice(T, t_overlap, replace=False):
                m[n * tt:n * (tt + 1)] = True
        masks.append(m)
    return torch.from_numpy(np.stack(masks)), torch.from_numpy(nfm), np.array(strat)


def affine_theta(angle, t1, t2, scale):
    """dalle_bert.py:168-202: [[s cos a, s sin(-a), t1], [s sin a, s cos a, t2]], in fp32 tensor arithmetic as the reference."""
    a, s = torch.tensor(angle, dtype=torch.float32), torch.tensor(scale, dtype=torch.float32)
    th = torch.empty(2, 3)
    th[0, 0], th[0, 1], th[0, 2] = s * torch.cos(a), s * torch.sin(-a), t1
    th[1, 0], th[1, 1], th[1, 2] = s * torch.sin(a), s * torch.cos(a), t2
    return th


def affine_warp(frame, theta):
    """frame [C,H,W]: F.affine_grid + F.grid_sample(padding_mode='reflection', align_corners=False) as line 197-201."""
    x = frame.unsqueeze(0)
    grid = F.affine_grid(theta.unsqueeze(0), x.size(), align_corners=False)
    return F.grid_sample(x, grid, padding_mode='reflection', align_corners=False)[0]


def apply_warp(x, params):
    """x [B,T,C,H,W]; params: list of dicts (mode, j1, src_b, src_t, chan, shift, theta, perm) -- dalle_bert.py:204-238."""
    out = x.clone()
    for b, p in enumerate(params):
        if p['mode'] == 0:
            out[b, p['j1']] = x[p['src_b'], p['src_t']]
        elif p['mode'] == 1:
            out[b] = x[b, p['perm']]
        elif p['mode'] == 2:
            m = torch.zeros_like(x[b, p['j1']])
            if p['chan'] == 0:
                m += p['shift']
            else:
                m[p['chan'] - 1] += p['shift']
            out[b, p['j1']] = torch.clamp(x[b, p['j1']] + m, 0, 1)
        else:
            out[b, p['j1']] = affine_warp(x[b, p['j1']], p['theta'])
    return out
