"""Host-side mirror of mmvid_pytorch/dalle_bert.py::BERT (259-1127) over the HIP kernels.

Same constructor, state_dict keys, attributes and methods as the reference class, so an MMVID-style
train.py / test.py drives it unchanged:
    forward(text, visual=, target=, return_loss=, rel=, vid=, ...) -> control_emb | (loss_msm, loss_rel, loss_vid)
    generate_images(text, visual=, mask_predict_steps=, mp_config=, ...) -> (images, pnag_samples, img_seq)
    get_image_tokens / recon_images / get_codebook_emb / decode_images / decode_masks / transformer_forward
What differs is where the arithmetic runs.  The three transformer passes of a training step (MSM, REL
negative, VID negative: dalle_bert.py:1037,1061,1101) are assembled as ONE batch of 3B sequences -- swapping
control embeddings along the batch (swap(), 110-122) is the same as swapping the control token ids -- and go
through the native tower once; sequence assembly, to_logits + cross-entropy and the whole backward are HIP
kernels.  Host-side stochastic choices (mask strategies 992-1029, warp 204-238) follow the reference's RNG
call order; tests inject them through the private `_mask1` / `_target_warp` kwargs.
"""
import random
from itertools import permutations

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .clip_tower import OpenAICLIPTransformer
from .functional import AssembleSequence, LNLinear, LNLinearCrossEntropy
from .modules import AxialPositionalEmbedding, AxialPositionalEmbeddingList
from .random_erasing import RandomErasing


def exists(val):
    return val is not None


def set_requires_grad(model, value):
    if model is not None:
        for p in model.parameters():
            p.requires_grad = value


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out

    return inner


class DivideMax(nn.Module):  # utils/utils.py:18-25 (only when stable=True; no driver enables it)
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        return x / x.amax(dim=self.dim, keepdim=True)


# ---- augmentation helpers (dalle_bert.py:93-238): host-side frame shuffles / colour / affine warps ----
PERM_LIST = None


def randperm(n, ordered=False):
    global PERM_LIST
    if ordered:
        return torch.randperm(n)
    if n < 6:
        if PERM_LIST is None:
            PERM_LIST = list(permutations(range(n)))[1:]
        return random.choice(PERM_LIST)
    perm_ord = torch.tensor(range(n))
    while True:
        perm = torch.randperm(n)
        if (perm != perm_ord).any():
            return perm


def swap(tensor, dim=0):
    if tensor.shape[dim] % 2 == 0:
        return torch.cat(torch.chunk(tensor, 2, dim=dim)[::-1], dim=dim)
    idx_perm = randperm(tensor.shape[dim], False)
    return tensor[idx_perm, ...] if dim == 0 else tensor[:, idx_perm, ...]


def warp_with_color(x):
    c_shift = (torch.rand(1) - 0.5).to(x.device)
    m = torch.zeros_like(x)
    num = random.randint(0, 3)
    if num == 0:
        m += c_shift
    else:
        m[num - 1] += c_shift
    return torch.clamp(x + m, 0, 1).unsqueeze(0)


def warp_video_with_color(video):
    out = []
    for n in range(video.shape[0]):
        x = video[n]
        c_shift = (torch.rand(1) - 0.5).to(x.device)
        m = torch.zeros_like(x)
        num = random.randint(0, 3)
        if num == 0:
            m += c_shift
        else:
            m[:, num - 1] += c_shift
        out.append(torch.clamp(x + m, 0, 1))
    return torch.stack(out)


def warp_with_affine(x, angle=180, trans=0.1, scale=0.05):
    angle = np.pi * angle / 180.
    pa = torch.FloatTensor(4)
    th = torch.FloatTensor(2, 3)
    pa[0].uniform_(-angle, angle)
    pa[1].uniform_(-trans, trans)
    pa[2].uniform_(-trans, trans)
    pa[3].uniform_(1. - scale, 1. + scale)
    th[0][0] = pa[3] * torch.cos(pa[0])
    th[0][1] = pa[3] * torch.sin(-pa[0])
    th[0][2] = pa[1]
    th[1][0] = pa[3] * torch.sin(pa[0])
    th[1][1] = pa[3] * torch.cos(pa[0])
    th[1][2] = pa[2]
    x = x.unsqueeze(0)
    grid = F.affine_grid(th.unsqueeze(0), x.size(), align_corners=False).to(x.device)
    return F.grid_sample(x, grid, padding_mode='reflection', align_corners=False)


def warp(x, vid_strategy_prob=(0.25, 0.25, 0.25, 0.25)):
    b, t, c, h, w = x.shape
    out = []
    for i in range(b):
        strategy = np.random.choice(range(4), p=vid_strategy_prob)
        if strategy == 0:  # a frame from another sequence
            i_ = np.random.choice(list(set(range(b)) - {i}))
            y = x[i].detach().clone()
            j1, j2 = random.randint(0, t - 1), random.randint(0, t - 1)
            y[j1, ...] = x[i_, j2, ...]
        elif strategy == 1:  # shuffle frames
            y = x[i, randperm(t), ...].detach().clone()
        elif strategy == 2:  # colour
            j1 = random.randint(0, t - 1)
            y = x[i].detach().clone()
            y[j1, ...] = warp_with_color(y[j1]).squeeze(0)
        else:  # affine
            j1 = random.randint(0, t - 1)
            y = x[i].detach().clone()
            y[j1, ...] = warp_with_affine(y[j1], 30, 0.1, 0.1).squeeze(0)
        out.append(y)
    return torch.stack(out, 0)


class BERT(nn.Module):
    def __init__(self, *, dim, vae, cvae=None, num_text_tokens=10000, text_seq_len=256, stable=False,
                 text_feature_dim=0, fixed_language_model=None, which_transformer='none', num_visuals=1,
                 num_targets=1, use_separate_visual_emb=False, insert_sep=False, text_emb_bottleneck=False,
                 **kwargs):
        super().__init__()
        if fixed_language_model is not None:
            raise NotImplementedError('fixed_language_model (RoBERTa features, dalle_bert.py:306-322) is outside '
                                      'the hot path of this build')
        image_size = vae.image_size
        num_image_tokens = vae.num_tokens
        image_fmap_size = vae.image_size // (2**vae.num_layers)
        image_seq_len = image_fmap_size**2
        self.dim, self.num_visuals, self.num_targets = dim, num_visuals, num_targets
        self.random_erasing = RandomErasing(p=1, scale=(0.2, 0.8), ratio=(0.5, 2), value=0)

        num_text_tokens = num_text_tokens + text_seq_len  # unique pad id per position (dalle_bert.py:299)
        self.text_emb = nn.Embedding(num_text_tokens, dim)
        self.text_pos_emb = nn.Embedding(text_seq_len, dim)
        self.image_emb = nn.Embedding(num_image_tokens + 2, dim)
        self.target_pos_emb = AxialPositionalEmbedding(dim, axial_shape=(num_targets, image_fmap_size, image_fmap_size))
        if cvae is not None:
            use_separate_visual_emb = True
        if num_visuals > 0:
            self.visual_emb = nn.Embedding(num_image_tokens + 2, dim) if use_separate_visual_emb else None
            self.visual_pos_emb = AxialPositionalEmbeddingList(dim, num_visuals,
                                                               axial_shape=(image_fmap_size, image_fmap_size))
        self.image_token_lut = {'[MASK]': num_image_tokens, '[SEP]': num_image_tokens + 1}
        self.num_text_tokens, self.num_image_tokens = num_text_tokens, num_image_tokens
        self.text_seq_len, self.image_seq_len = text_seq_len, image_seq_len
        self.image_fmap_size, self.image_size = image_fmap_size, image_size
        self.visual_seq_len = num_visuals * image_seq_len + (num_visuals * insert_sep)
        self.target_seq_len = num_targets * image_seq_len
        self.insert_sep = insert_sep
        self.special_token_lut = {'[REL]': 0, '[ST1]': 1, '[VID]': 2, '[ST3]': 3, '[ST4]': 4}
        self.num_special_tokens = len(self.special_token_lut)
        self.before_control_tok, self.after_control_tok = [0], [1, 2]
        self.before_control_seq_len, self.after_control_seq_len = 1, 2
        self.special_emb = nn.Embedding(self.num_special_tokens, dim)
        self.special_pos_emb = nn.Embedding(self.num_special_tokens, dim)
        self.rel_tok_index = 0
        self.st1_tok_index = 1 + self.text_seq_len + self.visual_seq_len
        self.vid_tok_index = self.st1_tok_index + 1
        self.txt_tok_index = 1
        self.control_seq_len = self.vid_tok_index + 1
        self.total_seq_len = self.control_seq_len + self.target_seq_len

        self.vae, self.cvae = vae, cvae
        set_requires_grad(self.vae, False)
        set_requires_grad(self.cvae, False)
        self.fixed_language_model = None
        self.which_transformer = which_transformer
        assert which_transformer != 'default'
        if not which_transformer.startswith('openai_clip'):
            raise NotImplementedError  # dalle_bert.py:406-407
        self.transformer = OpenAICLIPTransformer(self.total_seq_len, which_transformer,
                                                 model_path=kwargs.get('openai_clip_path'), causal=True,
                                                 mask_type='mask_prev',
                                                 mask_kwargs={'index': [self.st1_tok_index, self.vid_tok_index]},
                                                 layers=kwargs.get('transformer_layers'))
        self.stable = stable
        if stable:
            self.norm_by_max = DivideMax(dim=-1)
        self.to_logits = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, self.num_image_tokens))
        self.to_logits_rel = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, 1))
        self.to_logits_vid = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, 1))
        self.current_step = 0
        self.visual_eraser = RandomErasing(p=0.95, scale=(0.55, 0.85), ratio=(0.5, 2), value=self.num_image_tokens)
        self._w16_cache = {}
        # segment table: which embedding table each position reads (0 special, 1 text, 2 visual, 3 image)
        seg = [0] + [1] * self.text_seq_len + [2] * self.visual_seq_len + [0, 0] + [3] * self.target_seq_len
        self.register_buffer('_seg', torch.tensor(seg, dtype=torch.int32), persistent=False)
        self.register_buffer('_st_vid', torch.tensor([[1, 2]], dtype=torch.long), persistent=False)  # [ST1] [VID]

    # ------------------------------------------------------------------------------------ small helpers
    def _w16(self, lin):
        """bf16 shadow of a head's Linear weight (refreshed when the parameter changes)."""
        w = lin.weight
        key = (w._version, w.data_ptr())
        c = self._w16_cache.get(id(lin))
        if c is None or c[0] != key:
            c = (key, ops.cast_bf16(w.detach().contiguous()))
            self._w16_cache[id(lin)] = c
        return c[1]

    def attach_head_shadow(self, lin, view):
        """Engine hook (fused optimiser keeps `view` == bf16(weight))."""
        self._w16_cache[id(lin)] = ((lin.weight._version, lin.weight.data_ptr()), view)

    def _tables(self):
        vis = self.visual_emb.weight if (self.num_visuals > 0 and self.visual_emb is not None) else self.image_emb.weight
        return (self.special_emb.weight, self.text_emb.weight, vis, self.image_emb.weight)

    def _pos_table(self):
        """[total_seq_len, dim]: special_pos / text_pos / axial tables laid out along the sequence."""
        sp = self.special_pos_emb.weight
        parts = [sp[0:1], self.text_pos_emb.weight[:self.text_seq_len]]
        if self.num_visuals > 0:
            parts.append(self.visual_pos_emb.table(insert_sep=bool(self.insert_sep)))
        parts += [sp[1:3], self.target_pos_emb.table()]
        return torch.cat(parts, 0)

    def _small_head(self, seq, x):
        return F.linear(F.layer_norm(x, (self.dim, ), seq[0].weight, seq[0].bias, seq[0].eps), seq[1].weight, seq[1].bias)

    def get_special_token(self, tok_list, batch_size=1, device='cuda'):
        return torch.tensor(tok_list, dtype=torch.long, device=device).repeat(batch_size, 1)

    def transformer_forward(self, tokens):
        out = self.transformer(tokens)
        if self.stable:
            out = self.norm_by_max(out)
        return out

    def to_logits_rows(self, x):
        """to_logits on [..., dim] rows (HIP LayerNorm + MFMA GEMM)."""
        shp = x.shape
        lin = self.to_logits[1]
        y = LNLinear.apply(x.reshape(-1, shp[-1]), self.to_logits[0].weight, self.to_logits[0].bias, lin.weight,
                           lin.bias, self._w16(lin))
        return y.view(*shp[:-1], -1)

    # --------------------------------------------------------------------------------- token helpers
    def get_image_tokens(self, image, reshape=True, insert_sep=False, which_vae='vae'):
        vae = self.cvae if (which_vae == 'cvae' and self.cvae is not None) else self.vae
        if isinstance(image, list):
            assert len(image[0].shape) == 4, 'image should be list of 4d image tensors'
            image = torch.stack(image, dim=1)
        if len(image.shape) == 4:
            image = image.unsqueeze(1)
        if len(image.shape) == 5:
            b, t, c, h, w = image.shape
            s = vae.image_size
            assert (c, h, w) == (3, s, s), f'invalid image of dimensions {image.shape} passed in during training'
            image = vae.get_codebook_indices(image.reshape(b * t, c, h, w))
            if reshape:
                if insert_sep:
                    image = image.view(b, t, -1)
                    sep = torch.full((b, t, 1), self.image_token_lut['[SEP]'], device=image.device, dtype=torch.long)
                    image = torch.cat((image, sep), dim=2).reshape(b, -1)
                else:
                    image = image.view(b, -1)
        return image

    @torch.no_grad()
    def recon_images(self, images, which_vae='vae'):
        vae = self.cvae if (which_vae == 'cvae' and self.cvae is not None) else self.vae
        return vae.decode(self.get_image_tokens(images, reshape=False, which_vae=which_vae))

    @torch.no_grad()
    def get_codebook_emb(self, images, which_vae='vae'):
        b, t = images.shape[:2]
        img_seq = self.get_image_tokens(images, reshape=False, which_vae=which_vae)
        img_code = img_seq.view(b, t, -1)
        return img_code, ops.gather_rows(self.image_emb.weight.detach(), img_code.contiguous())

    def decode_images(self, img_seq):
        return self.vae.decode(img_seq.reshape(-1, self.image_seq_len))

    def decode_masks(self, mask):
        f = self.image_fmap_size
        mask = mask.reshape(-1, 1, f, f)
        patch = self.image_size // f
        mask_ = torch.repeat_interleave(torch.repeat_interleave(mask, patch, 2), patch, 3)
        return F.pad(mask_, (0, 0, 0, 0, 0, 2))

    def random_erase_codebook(self, image, eraser, erase_half=False):
        f = self.image_fmap_size
        image = image.view(image.shape[0], -1, f, f)
        if erase_half:
            image[:, :, f // 2:, :] = self.image_token_lut['[MASK]']
        else:
            image = torch.stack([eraser(c) for c in image], dim=0)
        return image.reshape(image.shape[0], -1)

    def erase_codebook_face(self, image, vc_mode, face_mode=None):
        f, M = self.image_fmap_size, self.image_token_lut['[MASK]']
        image = image.view(image.shape[0], -1, f, f)
        blank = torch.full_like(image, M)
        if vc_mode == 'face_8x8':
            if face_mode is None:
                face_mode = 'eyes_nose' if random.random() < 0.5 else 'mouth'
            if face_mode == 'eyes_nose':
                blank[:, :, 2:5, 1:7] = image[:, :, 2:5, 1:7]
            else:
                blank[:, :, 5:7, 2:6] = image[:, :, 5:7, 2:6]
            image = blank
        elif vc_mode == 'face2_8x8':
            blank[:, 0, ...] = image[:, 0, ...]
            blank[:, 1:, 2:6, 2:6] = image[:, 1:, 2:6, 2:6]
            image = blank
        elif vc_mode == 'face3_8x8':
            blank[:, 0, ...] = image[:, 0, ...]
            blank[:, :, 2:6, 2:6] = image[:, :, 2:6, 2:6]
            image = blank
        elif vc_mode in ('mask_8x8', 'mask2_8x8'):
            which = np.random.choice([1, 2, 3], p=[0.5, 0.25, 0.25]) if face_mode is None else 3
            if which == 2:
                blank[:, :, 2:6, 2:6] = image[:, :, 2:6, 2:6]
                image = blank
            elif which == 3:
                blank[:, :, 1:7, 1:7] = image[:, :, 1:7, 1:7]
                image = blank
        elif vc_mode == 'shape_4x4':
            image[:, :, 1:3, 1:3] = M
        else:
            raise NotImplementedError
        return image.reshape(image.shape[0], -1)

    # ------------------------------------------------------------------------------ sequence assembly
    def _control_ids(self, text, visual, erase_visual, erase_visual_half, vc_mode, face_mode, visual_aug_mode):
        """Token ids of [REL] text visual [ST1] [VID]  (dalle_bert.py:899-973) -> [B, control_seq_len] int64."""
        device, B = text.device, text.shape[0]
        assert text.shape[-1] == self.text_seq_len, \
            f'the length {text.shape[-1]} of the text tokens you passed in does not have the correct length ({self.text_seq_len})'
        text_range = torch.arange(self.text_seq_len, device=device) + (self.num_text_tokens - self.text_seq_len)
        text = torch.where(text == 0, text_range, text)
        parts = [torch.zeros(B, 1, dtype=torch.long, device=device), text]
        if self.num_visuals > 0:
            if exists(visual) and len(visual):
                if visual_aug_mode == 'motion_color' and random.random() < 0.9:
                    visual_ = visual.detach().clone()
                    visual_[:, 1:, ...] = warp_video_with_color(visual[:, 1:, ...])
                    visual = visual_
                visual = self.get_image_tokens(visual, insert_sep=self.insert_sep, which_vae='cvae')
                if erase_visual:
                    visual = self.random_erase_codebook(visual, self.visual_eraser, erase_visual_half)
                if vc_mode is not None:
                    visual = self.erase_codebook_face(visual, vc_mode, face_mode)
            else:
                visual = torch.full((B, self.visual_seq_len), self.image_token_lut['[MASK]'], dtype=torch.long, device=device)
            parts.append(visual)
        parts.append(self._st_vid.expand(B, 2))  # a buffer, not an upload: the step can be captured in a graph
        return torch.cat(parts, 1)

    def _assemble(self, ids, length):
        pos = self._pos_table()[:length]
        return AssembleSequence.apply(pos, ids.contiguous(), self._seg[:length].contiguous(), *self._tables())

    def _msm_mask(self, batch_size, device, msm_strategy_prob, msm_bernoulli_prob, pc_prob):
        """Masking strategies of dalle_bert.py:992-1029 (same RNG call order)."""
        f = self.image_fmap_size
        mask1_, nfm = [], torch.ones(batch_size, device=device)
        for i in range(batch_size):
            which = np.random.choice([1, 2, 3, 4], p=msm_strategy_prob)
            if which == 1:
                p = np.random.uniform(*msm_bernoulli_prob)
                mask1 = torch.bernoulli(torch.ones(self.target_seq_len, device=device) * p)
            elif which == 2:
                nfm[i] = 0
                mask1 = torch.zeros(self.target_seq_len, device=device)
            elif which == 3:
                mask1 = self.random_erasing(torch.ones(self.num_targets, 1, f, f, device=device)).reshape(-1)
            else:
                mask1 = 1 - self.random_erasing(torch.ones(self.num_targets, 1, f, f, device=device)).reshape(-1)
            if pc_prob > 0 and random.random() < pc_prob:
                t_overlap = random.randint(1, self.num_targets // 2)
                for tt in random.sample(range(self.num_targets), t_overlap):
                    mask1[self.image_seq_len * tt:self.image_seq_len * (tt + 1)] = 1
            mask1_.append(mask1)
        return torch.stack(mask1_, 0) == 1, nfm

    # ----------------------------------------------------------------------------------------- forward
    def forward(self, text, visual=None, target=None, mask=None, return_loss=False, rel=False, vid=False,
                erase_visual=False, erase_visual_half=False, msm_strategy_prob=[0.7, 0.1, 0.1, 0.1],
                msm_bernoulli_prob=[0.2, 0.5], rel_no_fully_masked=False,
                vid_strategy_prob=[0.25, 0.25, 0.25, 0.25], negvc=False, visual_neg=None, text_neg=None, pc_prob=0,
                vc_mode=None, face_mode=None, visual_aug_mode=None, _mask1=None, _target_warp=None, **kwargs):
        device = text.device
        B = text.shape[0]
        csl = self.control_seq_len
        ctrl_ids = self._control_ids(text, visual, erase_visual, erase_visual_half, vc_mode, face_mode, visual_aug_mode)
        if not return_loss:
            return self._assemble(ctrl_ids, csl)

        # Host-side stochastic choices first, in the reference's RNG order (mask strategies 992-1029, then warp 1094);
        # neither depends on device results, so both VQGAN encodes of the step can run as ONE batch of 2*B*T frames.
        if _mask1 is None:
            mask1, not_fully_masked = self._msm_mask(B, device, msm_strategy_prob, msm_bernoulli_prob, pc_prob)
        else:
            mask1 = _mask1.to(device)
            not_fully_masked = kwargs.get('_not_fully_masked', torch.ones(B, device=device))
        do_vid = vid and self.num_targets > 1
        target_warp = None
        if do_vid:
            target_warp = (_target_warp if _target_warp is not None else warp(target.detach(), vid_strategy_prob)).to(device)
        if do_vid and torch.is_tensor(target) and target.dim() == 5 and target_warp.shape == target.shape:
            toks = self.get_image_tokens(torch.cat((target, target_warp), 0))
            target, target_warp = toks[:B], toks[B:]
        else:
            target = self.get_image_tokens(target)
            if do_vid:
                target_warp = self.get_image_tokens(target_warp)
        MASK = self.image_token_lut['[MASK]']
        target_masked = torch.where(mask1, target, torch.full_like(target, MASK))

        # ---- the 1-3 sequences of this step as one batch
        seqs = [torch.cat((ctrl_ids, target_masked), 1)]
        if rel:
            assert B >= 2 and B % 2 == 0  # for REL swapping (dalle_bert.py:1045-1046)
            if negvc:
                if self.num_visuals > 0:
                    raise NotImplementedError('negvc with visuals: the reference builds a control_neg sequence without '
                                              'the visual segment (dalle_bert.py:923-930,974-975), whose length is inconsistent')
                neg_ids = self._control_ids(text_neg, None, False, False, None, None, None)
            else:
                neg_ids = swap(ctrl_ids, 0)
            seqs.append(torch.cat((neg_ids, target_masked), 1))
        if do_vid:
            warp_masked = torch.where(mask1, target_warp, torch.full_like(target_warp, MASK))
            seqs.append(torch.cat((ctrl_ids, warp_masked), 1))
        ids = torch.cat(seqs, 0)
        x = self._assemble(ids, self.total_seq_len)
        y = self.transformer_forward(x)  # [nB, L, dim]
        out = y[:B]

        # ---- MSM: to_logits + cross entropy over the masked positions (dalle_bert.py:1038-1040)
        lin = self.to_logits[1]
        rows = out[:, csl:, :].reshape(B * self.target_seq_len, self.dim)
        loss_msm, logits_msm = LNLinearCrossEntropy.apply(rows, target.reshape(-1).contiguous(), (~mask1).reshape(-1),
                                                          self.to_logits[0].weight, self.to_logits[0].bias, lin.weight,
                                                          lin.bias, self._w16(lin))
        self._last_logits_msm = logits_msm.view(B, self.target_seq_len, -1)

        nxt = 1
        if rel:
            out_neg = y[B * nxt:B * (nxt + 1)]
            nxt += 1
            lp = self._small_head(self.to_logits_rel, out[:, self.rel_tok_index, :]).squeeze()
            ln = self._small_head(self.to_logits_rel, out_neg[:, self.rel_tok_index, :]).squeeze()
            ones, zeros = torch.ones(B, device=device), torch.zeros(B, device=device)
            if rel_no_fully_masked:
                a = F.binary_cross_entropy_with_logits(lp, ones, reduction='none')
                b_ = F.binary_cross_entropy_with_logits(ln, zeros, reduction='none')
                loss_rel = (a * not_fully_masked + b_ * not_fully_masked).sum() / not_fully_masked.sum().clamp(min=1.)  # max(1., .) without a host sync
            else:
                loss_rel = F.binary_cross_entropy_with_logits(lp, ones) + F.binary_cross_entropy_with_logits(ln, zeros)
        else:
            loss_rel = torch.tensor(0.0, device=device)
        if do_vid:
            out_neg = y[B * nxt:B * (nxt + 1)]
            lp = self._small_head(self.to_logits_vid, out[:, self.vid_tok_index, :])
            ln = self._small_head(self.to_logits_vid, out_neg[:, self.vid_tok_index, :])
            ones, zeros = torch.ones(B, 1, device=device), torch.zeros(B, 1, device=device)
            if rel_no_fully_masked:  # NB: the reference does not weight by not_fully_masked here (1107-1116)
                den = not_fully_masked.sum().clamp(min=1.)
                loss_vid = F.binary_cross_entropy_with_logits(lp, ones, reduction='none').sum() / den + \
                    F.binary_cross_entropy_with_logits(ln, zeros, reduction='none').sum() / den
            else:
                loss_vid = F.binary_cross_entropy_with_logits(lp, ones) + F.binary_cross_entropy_with_logits(ln, zeros)
        else:
            loss_vid = torch.tensor(0.0, device=device)
        return loss_msm, loss_rel, loss_vid

    # ---------------------------------------------------------------------------------------- sampling
    @torch.no_grad()
    @eval_decorator
    def generate_images(self, text, *, visual=None, mask=None, img=None, argmax=False, dynamic=True, debug=False,
                        erase_visual=False, mask_predict_steps=10, preserve=None, t_overlap=1, pc_mode=None,
                        vc_mode=None, face_mode=None, mp_config=None, long_mode='long'):
        control_emb = self(text, visual=visual, erase_visual=erase_visual, erase_visual_half=True, vc_mode=vc_mode,
                           face_mode=face_mode, return_loss=False)
        img_seq, pnag_samples = self.mask_predict(control_emb, argmax=argmax, dynamic=dynamic, debug=debug,
                                                  steps=mask_predict_steps, preserve=preserve, t_overlap=t_overlap,
                                                  pc_mode=pc_mode, mp_config=mp_config, long_mode=long_mode)
        img_seq = img_seq.reshape(-1, self.image_seq_len)
        images = self.vae.decode(img_seq)
        images = images.view(-1, self.num_targets, *images.shape[1:])
        return images, pnag_samples, img_seq

    @torch.no_grad()
    def mask_predict(self, control_emb, dynamic=True, debug=False, steps=10, preserve=None, t_overlap=1,
                     mp_config=None, long_mode='long', **kwargs):
        """Mask-predict sampler, dalle_bert.py:514-714 (same schedule, same sampling rule)."""
        def sample_gumbel(logit, eps=1e-20):
            U = torch.rand_like(logit)
            return -torch.log(-torch.log(U + eps) + eps)

        def sample_multinomial(logits, temperature=1.):
            logits = logits + temperature * sample_gumbel(logits)
            probs = F.softmax(logits, dim=2)
            b, n, c = probs.shape
            tok = torch.multinomial(probs.reshape(b * n, c), 1).view(b, n, 1)
            Y = torch.gather(probs, 2, tok)
            return Y.squeeze(2), tok.squeeze(2)

        csl, device = control_emb.shape[1], control_emb.device
        TS, ISL, MASK = self.target_seq_len, self.image_seq_len, self.image_token_lut['[MASK]']
        interp = long_mode in ('interp', 'interp2', 'interp_real')
        if long_mode == 'long':
            if preserve is None:
                t_overlap = 0
            N = TS - ISL * t_overlap
        elif interp:
            N = TS // 2
        else:
            N = TS
        fully_masked_tok = torch.full((1, TS), MASK, dtype=torch.long, device=device)
        preserve_mask1 = torch.zeros(1, TS, dtype=torch.long, device=device)
        preserve_ = torch.full((control_emb.shape[0], TS), MASK, dtype=torch.long, device=device)
        if preserve is not None:
            if long_mode == 'long':
                preserve_mask1[:, :ISL * t_overlap] = 1
                preserve = preserve.reshape(-1, self.num_targets * preserve.shape[-1])
                preserve_[:, :ISL * t_overlap] = preserve[:, -ISL * t_overlap:]
            elif interp:
                pm = preserve_mask1.view(1, self.num_targets, ISL)
                pm[:, ::2, :] = 1
                pv = preserve.reshape(preserve.shape[0], self.num_targets, ISL)
                pr = preserve_.view(-1, self.num_targets, ISL)
                pr[:, ::2, :] = pv[:, :self.num_targets // 2, :]
        no_preserve = preserve is None
        preserve = preserve_
        preserve_mask1 = preserve_mask1 == 1

        iemb = self.image_emb.weight.detach()
        target_pos_emb = self.target_pos_emb.table().unsqueeze(0)
        mask_emb = iemb[MASK]

        Tmax = mp_config['T'] if steps <= 0 else steps
        Bm = mp_config['B']
        c = mp_config
        N3_n, N4_n = max(1, int(N * c['N3_n'])), max(1, int(N * c['N4_n']))
        n = list(N * np.linspace(c['N1_n'], c['N2_n'], c['T1_n'])) + list(N3_n * np.ones(c['T2_n'])) + \
            list(N4_n * np.ones(c['T3_n']))
        temp = list(np.linspace(c['N1_t'], c['N2_t'], c['T1_t'])) + list(c['N3_t'] * np.ones(c['T2_t'])) + \
            list(c['N4_t'] * np.ones(c['T3_t']))
        n = list(map(int, n))

        def run(emb_in):
            tokens = torch.cat((control_emb_, emb_in + target_pos_emb), dim=1)
            return self.transformer_forward(tokens)

        sample_toks, image_samples = [], []
        for i in range(control_emb.shape[0]):
            control_emb_ = control_emb[i:i + 1, ...]
            tok_in = fully_masked_tok
            if not no_preserve:
                tok_in[0, ...] = torch.where(preserve_mask1[0], preserve[i, ...], fully_masked_tok[0, ...])
            out = run(ops.gather_rows(iemb, tok_in))[:, csl:, :]
            Y, I_new = sample_multinomial(self.to_logits_rows(out), temp[0])
            I_tok = torch.where(preserve_mask1, preserve[i:i + 1, ...], I_new)
            if debug:
                image_samples.append(self.decode_images(I_tok))
            Smax, tmax, Imax = 0, 0, None
            for t in range(1, Tmax):
                emb_in, masks1 = [], []
                for j in range(Bm):
                    Y_valid = Y[~preserve_mask1]
                    idx_valid = torch.arange(TS, device=device)[~preserve_mask1[0]]
                    try:
                        mask1_idx = torch.multinomial(Y_valid, N - n[t - 1], replacement=False)
                    except RuntimeError:
                        mask1_idx = torch.multinomial(Y_valid, 1, replacement=False)
                    mask1_idx = idx_valid[mask1_idx]
                    mask1 = torch.zeros(TS, device=device).scatter_(0, mask1_idx, 1).unsqueeze(0)
                    mask1[preserve_mask1] = 1
                    mask1 = mask1 == 1
                    masks1.append(mask1)
                    emb_in.append(torch.where(mask1.unsqueeze(2), ops.gather_rows(iemb, I_tok), mask_emb))
                S = torch.zeros(Bm)
                YB, tokB = [], []
                for j in range(Bm):
                    out = run(emb_in[j])
                    Y_new, I_new = sample_multinomial(self.to_logits_rows(out[:, csl:, :]), temp[t])
                    mask1_j = torch.bitwise_or(masks1[j], preserve_mask1)
                    Y = torch.where(mask1_j, Y, Y_new)
                    I_tok = torch.where(mask1_j, I_tok, I_new)
                    s_rel = torch.sigmoid(self._small_head(self.to_logits_rel, out[:, self.rel_tok_index, :]))
                    s_vid = torch.sigmoid(self._small_head(self.to_logits_vid, out[:, self.vid_tok_index, :]))
                    S[j] = (s_rel * 0.5 + s_vid * 0.5).item()
                    YB.append(Y)
                    tokB.append(I_tok)
                jmax = S.argmax()
                Y, I_tok = YB[jmax], tokB[jmax]
                if debug:
                    mask_img = self.decode_masks((~masks1[jmax]).float())
                    image_samples.append(torch.clamp(image_samples[-1] * 0.7 + mask_img * 0.4, 0, 1))
                    image_samples.append(self.decode_images(I_tok))
                if dynamic:
                    if S[jmax] > Smax:
                        tmax, Smax, Imax = t, S[jmax], I_tok
                    if t - tmax >= 5:
                        break
                else:
                    Imax = I_tok
            sample_toks.append(Imax)
        return torch.cat(sample_toks, 0), image_samples
