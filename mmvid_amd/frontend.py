"""Device-side stochastic front-end of a training step (csrc/frontend.hip): MSM masking strategies
(mmvid_pytorch/dalle_bert.py:992-1029), the VID negative `warp()` (204-238), visual-token erasing (779-848).

One counter-based generator keyed by (seed, step, sample, purpose): `Frontend.step` is a device scalar advanced by a
device op, so a captured training step draws fresh decisions on every replay with no host involvement.  Ranks use
different seeds (train.py:87 seeds every rank with seed + rank)."""
import ctypes

import torch

from . import _lib, ops
from .ops import _p, _stream

f32, u8, i64 = torch.float32, torch.uint8, torch.int64


def _farr(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


class Frontend:
    def __init__(self, seed=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.step = None  # device fp32 [1], created on first use (counts forward calls)
        self._warp_scratch = None

    def _step(self, device):
        dev = torch.device(device)
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        if self.step is None or self.step.device != dev:
            self.step = torch.zeros(1, device=dev, dtype=f32)
        return self.step

    def advance(self, device):
        """One more forward call happened: later draws use a new counter block (a device op: graph-capturable)."""
        _lib.call('mmvid_counter_add', _p(self._step(device)), 1.0, _stream())

    def msm_masks(self, B, T, fmap, device, strategy_prob, bernoulli_prob, pc_prob=0.0, want_strategy=False):
        """-> (mask1 uint8 [B, T*fmap*fmap] (1 = visible), not_fully_masked f32 [B][, strategy int32 [B]])."""
        mask1 = torch.empty(B, T * fmap * fmap, device=device, dtype=u8)
        nfm = torch.empty(B, device=device, dtype=f32)
        strat = torch.empty(B, device=device, dtype=torch.int32) if want_strategy else None
        p = list(strategy_prob)
        _lib.call('mmvid_msm_masks', self.seed, _p(self._step(device)), B, T, fmap, _farr(p), float(bernoulli_prob[0]),
                  float(bernoulli_prob[1]), float(pc_prob), _p(mask1), _p(nfm), _p(strat), _stream())
        return (mask1, nfm, strat) if want_strategy else (mask1, nfm)

    def vid_warp(self, x, strategy_prob, out=None, params=None):
        """x [B,T,C,H,W] f32 in [0,1] -> the VID negative.  `params` (a uint8 tensor of B * warp_params_bytes): apply
        these instead of drawing (tests)."""
        ops._chk(x, f32, 'x')
        B, T, C, H, W = x.shape
        nbytes = B * _lib.load().mmvid_warp_params_bytes()
        if params is None:
            if self._warp_scratch is None or self._warp_scratch.numel() < nbytes or self._warp_scratch.device != x.device:
                self._warp_scratch = torch.empty(nbytes, device=x.device, dtype=u8)
            scratch, draw = self._warp_scratch, 1
        else:
            scratch, draw = params, 0
        if out is None:
            out = torch.empty_like(x)
        _lib.call('mmvid_vid_warp', self.seed, _p(self._step(x.device)), _p(x), B, T, C, H, W, _farr(list(strategy_prob)),
                  _p(scratch), draw, _p(out), _stream())
        return out

    def _scratch(self, B, device):
        nbytes = B * _lib.load().mmvid_warp_params_bytes()
        if self._warp_scratch is None or self._warp_scratch.numel() < nbytes or self._warp_scratch.device != torch.device(device):
            self._warp_scratch = torch.empty(nbytes, device=device, dtype=u8)
        return self._warp_scratch

    def vid_warp_new_frames(self, x, strategy_prob, out):
        """Draw the warp parameters and write the ONE frame per sample whose pixels are new into out [B,C,H,W] (the rest
        of the negative is frames the VQGAN has already tokenised: vid_warp_tokens)."""
        ops._chk(x, f32, 'x')
        B, T, C, H, W = x.shape
        assert out.shape == (B, C, H, W) and out.is_contiguous() and out.dtype == f32
        _lib.call('mmvid_vid_warp_new_frames', self.seed, _p(self._step(x.device)), _p(x), B, T, C, H, W,
                  _farr(list(strategy_prob)), _p(self._scratch(B, x.device)), 1, _p(out), _stream())
        return out

    def vid_warp_tokens(self, target_tok, new_tok, T):
        """target_tok [B, T*n], new_tok [B, n] -> tokens of the negative drawn by the last vid_warp_new_frames call."""
        B, n = new_tok.shape
        out = torch.empty_like(target_tok)
        _lib.call('mmvid_vid_warp_tokens', _p(target_tok), _p(new_tok.contiguous()), _p(self._warp_scratch), B, T, n, _p(out),
                  _stream())
        return out

    def erase_choice(self, tok, Tv, fmap, value, choices, frame0_full=False):
        """tok [B, Tv*fmap*fmap] int64 (modified in place).  choices: list of (prob, mode, (r0, r1, c0, c1)); one is
        drawn per call.  mode 0 untouched | 1 keep only the box | 2 erase the box."""
        ops._chk(tok, i64, 'tok')
        n = len(choices)
        cum, acc = [], 0.0
        for pr, _, _ in choices:
            acc += pr
            cum.append(acc)
        cum[-1] = 2.0  # the last alternative absorbs rounding
        modes = (ctypes.c_int32 * n)(*[m for _, m, _ in choices])
        boxes = (ctypes.c_int32 * (4 * n))(*[v for _, _, bx in choices for v in bx])
        B = tok.shape[0]
        _lib.call('mmvid_erase_tokens_choice', self.seed, _p(self._step(tok.device)), n, _farr(cum), modes, boxes,
                  int(frame0_full), B, Tv, fmap, int(value), _p(tok), _stream())
        return tok

    def random_erase(self, tok, Tv, fmap, value, p, scale, ratio, erase_half=False):
        ops._chk(tok, i64, 'tok')
        _lib.call('mmvid_random_erase_tokens', self.seed, _p(self._step(tok.device)), tok.shape[0], Tv, fmap, float(p),
                  float(scale[0]), float(scale[1]), float(ratio[0]), float(ratio[1]), int(erase_half), int(value), _p(tok),
                  _stream())
        return tok


def face_choices(vc_mode, face_mode):
    """The token-map regions of erase_codebook_face (dalle_bert.py:796-848 / dalle_artv.py:356-416) as
    (alternatives, frame0_full).  Each alternative = (probability, mode, box); see Frontend.erase_choice."""
    whole = (0, 0, 0, 0)
    if vc_mode == 'face_8x8':  # only the eyes+nose or the mouth region survives
        eyes, mouth = (0.5, 1, (2, 5, 1, 7)), (0.5, 1, (5, 7, 2, 6))
        if face_mode is None:
            return [eyes, mouth], False
        return [(1.0, ) + (eyes[1:] if face_mode == 'eyes_nose' else mouth[1:])], False
    if vc_mode in ('face2_8x8', 'face3_8x8'):  # frame 0 whole, later frames only the centre
        return [(1.0, 1, (2, 6, 2, 6))], True
    if vc_mode in ('mask_8x8', 'mask2_8x8'):  # nothing / centre 4x4 / centre 6x6
        if face_mode is None:
            return [(0.5, 0, whole), (0.25, 1, (2, 6, 2, 6)), (0.25, 1, (1, 7, 1, 7))], False
        return [(1.0, 1, (1, 7, 1, 7))], False
    if vc_mode == 'shape_4x4':
        return [(1.0, 2, (1, 3, 1, 3))], False
    raise NotImplementedError(vc_mode)
