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
    """`state` is a device buffer of four 32-bit words: [0] the forward-call counter (fp32, advanced by a device op), [1..2]
    the 64-bit seed, [3] reserved.  The kernels read the seed from there (their by-value seed argument is XORed on top and
    is 0 here), so a step captured in a hipGraph follows `frontend.seed = ...` and `load_state_dict` without re-capture.
    seed=None (the default): derived at first use from torch.initial_seed() + rank, as train.py:87 seeds every rank with
    seed + rank -- ranks draw different masks / warps, and `torch.manual_seed` controls the stream."""

    def __init__(self, seed=None):
        self._seed = None if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self.state = None  # created on first use, on the device of that use
        self._pending_step = None
        self._warp_scratch = None

    # ---- seed / step live on the device once `state` exists
    @staticmethod
    def _default_seed():
        import torch.distributed as dist
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        return (int(torch.initial_seed()) + rank) & 0xFFFFFFFFFFFFFFFF

    @property
    def seed(self):
        if self._seed is None:
            self._seed = self._default_seed()
        return self._seed

    @seed.setter
    def seed(self, value):
        self._seed = int(value) & 0xFFFFFFFFFFFFFFFF
        if self.state is not None:
            self._write_seed()

    def _write_seed(self):
        lo, hi = self.seed & 0xFFFFFFFF, self.seed >> 32
        words = torch.tensor([lo - (1 << 32) if lo >= (1 << 31) else lo, hi - (1 << 32) if hi >= (1 << 31) else hi], dtype=torch.int32)
        self.state.view(torch.int32)[1:3].copy_(words)

    @property
    def step(self):
        """Device fp32 [1] view of the forward-call counter (None before the first use)."""
        return None if self.state is None else self.state[0:1]

    @step.setter
    def step(self, value):
        if value is None:
            self.state = None  # next use starts a fresh counter
        else:
            self._state(value.device)[0:1].copy_(value)

    def _state(self, device):
        dev = torch.device(device)
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        if self.state is None or self.state.device != dev:
            old = self.state
            self.state = torch.zeros(4, device=dev, dtype=f32)
            if old is not None:
                self.state[0:1].copy_(old[0:1])
            if self._pending_step is not None:  # load_state_dict before the first use
                self.state[0:1].fill_(self._pending_step)
                self._pending_step = None
            self._write_seed()
        return self.state

    def _step(self, device):
        return self._state(device)

    def state_dict(self):
        return {'seed': self.seed, 'step': 0.0 if self.state is None else float(self.state[0])}

    def load_state_dict(self, sd):
        self._seed = int(sd['seed']) & 0xFFFFFFFFFFFFFFFF
        self._pending_step = float(sd.get('step', 0.0))
        if self.state is not None:
            self.state[0:1].fill_(self._pending_step)
            self._write_seed()
            self._pending_step = None

    def advance(self, device):
        """One more forward call happened: later draws use a new counter block (a device op: graph-capturable)."""
        _lib.call('mmvid_counter_add', _p(self._step(device)), 1.0, _stream())

    def msm_masks(self, B, T, fmap, device, strategy_prob, bernoulli_prob, pc_prob=0.0, want_strategy=False):
        """-> (mask1 uint8 [B, T*fmap*fmap] (1 = visible), not_fully_masked f32 [B][, strategy int32 [B]])."""
        mask1 = torch.empty(B, T * fmap * fmap, device=device, dtype=u8)
        nfm = torch.empty(B, device=device, dtype=f32)
        strat = torch.empty(B, device=device, dtype=torch.int32) if want_strategy else None
        p = list(strategy_prob)
        _lib.call('mmvid_msm_masks', 0, _p(self._step(device)), B, T, fmap, _farr(p), float(bernoulli_prob[0]),
                  float(bernoulli_prob[1]), float(pc_prob), _p(mask1), _p(nfm), _p(strat), _stream())
        return (mask1, nfm, strat) if want_strategy else (mask1, nfm)

    @staticmethod
    def msm_masks_from_decisions(decisions, bernoulli, T, fmap):
        """The mask kernel on supplied decisions (tests): decisions int32 [B, 72], bernoulli uint8 [B, T*fmap*fmap] or None."""
        B = decisions.shape[0]
        mask1 = torch.empty(B, T * fmap * fmap, device=decisions.device, dtype=u8)
        nfm = torch.empty(B, device=decisions.device, dtype=f32)
        _lib.call('mmvid_msm_masks_inject', _p(ops._chk(decisions, torch.int32, 'decisions')), _p(bernoulli), B, T, fmap, _p(mask1),
                  _p(nfm), _stream())
        return mask1, nfm

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
        _lib.call('mmvid_vid_warp', 0, _p(self._step(x.device)), _p(x), B, T, C, H, W, _farr(list(strategy_prob)),
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
        _lib.call('mmvid_vid_warp_new_frames', 0, _p(self._step(x.device)), _p(x), B, T, C, H, W,
                  _farr(list(strategy_prob)), _p(self._scratch(B, x.device)), 1, _p(out), _stream())
        return out

    def vid_warp_tokens(self, target_tok, new_tok, T):
        """target_tok [B, T*n], new_tok [B, n] -> tokens of the negative drawn by the last vid_warp_new_frames call."""
        B, n = new_tok.shape
        out = torch.empty_like(target_tok)
        _lib.call('mmvid_vid_warp_tokens', _p(target_tok), _p(new_tok.contiguous()), _p(self._warp_scratch), B, T, n, _p(out),
                  _stream())
        return out

    def visual_color_jitter(self, visual, p=0.9, first_frame=1, want_params=False):
        """visual_aug_mode='motion_color': visual [B,Tv,C,H,W] -> a copy whose frames first_frame.. carry a per-sample colour
        shift with probability p (one gate per call)."""
        x = ops._chk(visual.detach().float().clone().contiguous(), f32, 'visual')
        B, Tv, C, H, W = x.shape
        params = torch.empty(B, 3, device=x.device, dtype=f32) if want_params else None
        _lib.call('mmvid_visual_color_jitter', 0, _p(self._step(x.device)), _p(x), B, Tv, C, H, W, float(p), int(first_frame),
                  _p(params), _stream())
        return (x, params) if want_params else x

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
        _lib.call('mmvid_erase_tokens_choice', 0, _p(self._step(tok.device)), n, _farr(cum), modes, boxes,
                  int(frame0_full), B, Tv, fmap, int(value), _p(tok), _stream())
        return tok

    def random_erase(self, tok, Tv, fmap, value, p, scale, ratio, erase_half=False):
        ops._chk(tok, i64, 'tok')
        _lib.call('mmvid_random_erase_tokens', 0, _p(self._step(tok.device)), tok.shape[0], Tv, fmap, float(p),
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
