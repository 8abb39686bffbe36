"""Host-side mirror of mmvid_pytorch/dalle_artv.py::DALLE (103-542), the autoregressive "ART-V" baseline, over
the HIP kernels: same constructor, state_dict keys, `forward` (logits or (loss, 0, 0)) and `generate_images`
-> (images, [], None).  The tower runs with the causal mask predicate; the 51,584-way `to_logits` is the bf16
MFMA GEMM.  generate_images follows the reference's loop (one forward per generated token, dalle_artv.py:
253-281) but evaluates `to_logits` on the last position only -- the only row the loop reads."""
import random

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .clip_tower import OpenAICLIPTransformer
from .dalle_bert import (DivideMax, eval_decorator, exists, set_requires_grad, warp_video_with_color)
from .functional import AssembleSequence, LNLinear
from .modules import AxialPositionalEmbedding, AxialPositionalEmbeddingList
from .random_erasing import RandomErasing


def is_empty(t):
    return t.nelement() == 0


def top_k(logits, thres=0.5):
    """dalle_artv.py:61-67."""
    k = max(int((1 - thres) * logits.shape[-1]), 1)
    val, ind = torch.topk(logits, k)
    probs = torch.full_like(logits, float('-inf'))
    probs.scatter_(1, ind, val)
    return probs


class DALLE(nn.Module):
    def __init__(self, *, dim, vae, cvae=None, num_text_tokens=10000, text_seq_len=256, loss_img_weight=7,
                 stable=False, which_transformer='none', num_visuals=1, num_targets=1, **kwargs):
        super().__init__()
        assert num_visuals > 0
        image_fmap_size = vae.image_size // (2**vae.num_layers)
        image_seq_len = image_fmap_size**2
        num_image_tokens = vae.num_tokens
        self.dim = dim
        self.target_seq_len = image_seq_len * num_targets
        self.visual_seq_len = image_seq_len * num_visuals
        self.control_seq_len = text_seq_len + self.visual_seq_len
        self.insert_sep = False
        num_text_tokens = num_text_tokens + text_seq_len
        num_visual_tokens = num_image_tokens + self.visual_seq_len
        self.text_emb = nn.Embedding(num_text_tokens, dim)
        self.image_emb = nn.Embedding(num_image_tokens, dim)
        self.text_pos_emb = nn.Embedding(text_seq_len + 1, dim)  # +1 for <bos>
        shape = (image_fmap_size, image_fmap_size) if num_targets == 1 else (num_targets, image_fmap_size, image_fmap_size)
        self.image_pos_emb = AxialPositionalEmbedding(dim, axial_shape=shape)
        self.visual_emb = nn.Embedding(num_visual_tokens, dim)
        self.visual_pos_emb = AxialPositionalEmbeddingList(dim, num_visuals, axial_shape=(image_fmap_size, image_fmap_size))
        self.num_text_tokens, self.num_image_tokens = num_text_tokens, num_image_tokens
        self.num_visual_tokens = num_visual_tokens
        self.num_control_tokens = num_text_tokens + num_visual_tokens
        self.text_seq_len, self.image_seq_len = text_seq_len, image_seq_len
        self.num_visuals, self.num_targets, self.image_fmap_size = num_visuals, num_targets, image_fmap_size
        self.special_token_lut = {'[REL]': 0, '[ST1]': 1, '[ST2]': 2, '[ST3]': 3}
        self.num_special_tokens, self.num_estimation_tokens = 4, 2
        self.special_emb = nn.Embedding(self.num_special_tokens, dim)        # unused in forward (as the reference)
        self.estimation_pos_emb = nn.Embedding(self.num_estimation_tokens, dim)  # unused in forward
        self.total_tokens = num_text_tokens + num_image_tokens + num_visual_tokens
        self.total_seq_len = text_seq_len + self.target_seq_len + self.visual_seq_len
        self.vae, self.cvae = vae, cvae
        set_requires_grad(self.vae, False)
        set_requires_grad(self.cvae, False)
        self.which_transformer = which_transformer
        if not which_transformer.startswith('openai_clip'):
            raise NotImplementedError
        self.transformer = OpenAICLIPTransformer(self.total_seq_len, which_transformer,
                                                 model_path=kwargs.get('openai_clip_path'),
                                                 layers=kwargs.get('transformer_layers'))
        self.stable = stable
        if stable:
            self.norm_by_max = DivideMax(dim=-1)
        self.to_logits = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, self.total_tokens))
        # block-diagonal vocabulary mask (dalle_artv.py:215-227) kept as three (row range -> column range) segments
        self.loss_vis_weight, self.loss_img_weight = 1., loss_img_weight
        self.eraser = RandomErasing(p=1, scale=(0.4, 0.8), ratio=(0.5, 2), value=-1)
        self._w16_cache = None
        seg = [0] * (text_seq_len + 1) + [1] * self.visual_seq_len + [2] * self.target_seq_len
        self.register_buffer('_seg', torch.tensor(seg, dtype=torch.int32), persistent=False)

    @property
    def logits_mask(self):
        """The reference's [1, total_seq_len, total_tokens] bool buffer, materialised on demand (tests only)."""
        m = torch.block_diag(torch.ones(self.text_seq_len, self.num_text_tokens),
                             torch.ones(self.visual_seq_len, self.num_visual_tokens),
                             torch.ones(self.target_seq_len, self.num_image_tokens)) == 0
        return m.unsqueeze(0)

    def _w16(self):
        w = self.to_logits[1].weight
        key = (w._version, w.data_ptr())
        if self._w16_cache is None or self._w16_cache[0] != key:
            self._w16_cache = (key, ops.cast_bf16(w.detach().contiguous()))
        return self._w16_cache[1]

    def _allowed_range(self, pos):
        """columns of the logits that position `pos` may predict (block diagonal, dalle_artv.py:215-219)."""
        if pos < self.text_seq_len:
            return 0, self.num_text_tokens
        if pos < self.control_seq_len:
            return self.num_text_tokens, self.num_control_tokens
        return self.num_control_tokens, self.total_tokens

    # token helpers shared with BERT's behaviour (dalle_artv.py:306-416)
    def get_image_tokens(self, image, reshape=True, insert_sep=False, which_vae='vae'):
        vae = self.cvae if (which_vae == 'cvae' and self.cvae is not None) else self.vae
        if isinstance(image, list):
            image = torch.stack(image, dim=1)
        if len(image.shape) == 4:
            image = image.unsqueeze(1)
        if len(image.shape) == 5:
            b, t, c, h, w = image.shape
            s = vae.image_size
            assert (c, h, w) == (3, s, s), f'invalid image of dimensions {image.shape} passed in during training'
            image = vae.get_codebook_indices(image.reshape(b * t, c, h, w))
            if reshape:
                image = image.view(b, -1)
        return image

    def random_erase_codebook(self, image, eraser, erase_half=False):
        f = self.image_fmap_size
        image = image.view(image.shape[0], -1, f, f)
        if erase_half:
            image[:, :, f // 2:, :] = -1
        else:
            image = torch.stack([eraser(c) for c in image], dim=0)
        return image.reshape(image.shape[0], -1)

    def _hidden(self, text, visual, image, erase_visual, erase_visual_half, vc_mode, face_mode, visual_aug_mode):
        """dalle_artv.py:431-500: ids -> assembled sequence -> causal tower.  Returns (out, labels parts)."""
        assert text.shape[-1] == self.text_seq_len, \
            f'the length {text.shape[-1]} of the text tokens you passed in does not have the correct length ({self.text_seq_len})'
        device, B = text.device, text.shape[0]
        text_range = torch.arange(self.text_seq_len, device=device) + (self.num_text_tokens - self.text_seq_len)
        text = F.pad(torch.where(text == 0, text_range, text), (1, 0), value=0)  # <bos>
        if exists(visual) and not is_empty(visual):
            if visual_aug_mode == 'motion_color' and random.random() < 0.9:
                visual_ = visual.detach().clone()
                visual_[:, 1:, ...] = warp_video_with_color(visual[:, 1:, ...])
                visual = visual_
            visual = self.get_image_tokens(visual, which_vae='cvae')
            if erase_visual:
                visual = self.random_erase_codebook(visual, self.eraser, erase_visual_half)
            if vc_mode is not None:
                raise NotImplementedError('erase_codebook_face for ART-V (dalle_artv.py:356-416) is not on the benchmarked path')
        else:
            visual = -torch.ones(B, self.visual_seq_len, device=device).long()
        visual_range = torch.arange(self.visual_seq_len, device=device) + (self.num_visual_tokens - self.visual_seq_len)
        visual = torch.where(visual == -1, visual_range, visual)
        parts = [text, visual]
        if exists(image) and not is_empty(image):
            image = self.get_image_tokens(image)
            parts.append(image)
        ids = torch.cat(parts, 1)
        if ids.shape[1] > self.total_seq_len:  # drop the last token when training (dalle_artv.py:496-498)
            ids = ids[:, :-1]
        L = ids.shape[1]
        pos = torch.cat([self.text_pos_emb.weight, self.visual_pos_emb.table(), self.image_pos_emb.table()], 0)[:L]
        x = AssembleSequence.apply(pos, ids.contiguous(), self._seg[:L].contiguous(), self.text_emb.weight,
                                   self.visual_emb.weight, self.image_emb.weight)
        out = self.transformer(x)
        if self.stable:
            out = self.norm_by_max(out)
        return out, text, visual, image

    def _logits_rows(self, rows):
        lin = self.to_logits[1]
        return LNLinear.apply(rows, self.to_logits[0].weight, self.to_logits[0].bias, lin.weight, lin.bias, self._w16())

    def forward(self, text, visual=None, target=None, return_loss=False, erase_visual=False, erase_visual_half=False,
                vc_mode=None, face_mode=None, visual_aug_mode=None, **kwargs):
        out, text, visual, image = self._hidden(text, visual, target, erase_visual, erase_visual_half, vc_mode,
                                                face_mode, visual_aug_mode)
        B, L, _ = out.shape
        logits = self._logits_rows(out.reshape(B * L, self.dim)).view(B, L, -1)
        neg = -torch.finfo(logits.dtype).max
        # block-diagonal mask (dalle_artv.py:509-512), applied per segment instead of through a [L, V] bool buffer
        masked = torch.full_like(logits, neg)
        for lo, hi in ((0, self.text_seq_len), (self.text_seq_len, self.control_seq_len), (self.control_seq_len, L)):
            if lo < min(hi, L):
                c0, c1 = self._allowed_range(lo)
                masked[:, lo:min(hi, L), c0:c1] = logits[:, lo:min(hi, L), c0:c1]
        logits = masked
        if not return_loss:
            return logits
        assert exists(image), 'when training, image must be supplied'
        labels = torch.cat((text[:, 1:], visual + self.num_text_tokens, image + self.num_control_tokens), dim=1)
        lg = logits.permute(0, 2, 1)
        tl, cl = self.text_seq_len, self.control_seq_len
        loss_text = F.cross_entropy(lg[:, :, :tl], labels[:, :tl])
        loss_vis = F.cross_entropy(lg[:, :, tl:cl], labels[:, tl:cl])
        loss_img = F.cross_entropy(lg[:, :, cl:], labels[:, cl:])
        loss = (loss_text + self.loss_vis_weight * loss_vis + self.loss_img_weight * loss_img) / \
            (self.loss_img_weight + self.loss_vis_weight + 1)
        zero = torch.tensor(0.0, device=text.device)
        return loss, zero, zero

    def _embed_rows(self, ids, first_pos):
        """Embedding + positional rows for `ids` [B, n] occupying positions first_pos .. first_pos+n-1."""
        n = ids.shape[1]
        pos = torch.cat([self.text_pos_emb.weight, self.visual_pos_emb.table(), self.image_pos_emb.table()], 0)
        return ops.assemble_sequence([self.text_emb.weight, self.visual_emb.weight, self.image_emb.weight], ids.contiguous(),
                                     self._seg[first_pos:first_pos + n].contiguous(), pos[first_pos:first_pos + n].contiguous())

    def _sample(self, last_hidden, position, filter_thres, temperature):
        """dalle_artv.py:282-291: logits of the last position restricted to its allowed block, top-k, multinomial."""
        last = self._logits_rows(last_hidden.contiguous())
        c0, c1 = self._allowed_range(position)
        logits = torch.full_like(last, -torch.finfo(torch.float32).max)
        logits[:, c0:c1] = last[:, c0:c1]
        probs = F.softmax(top_k(logits, thres=filter_thres) / temperature, dim=-1)
        return torch.multinomial(probs, 1) - self.num_control_tokens

    @torch.no_grad()
    @eval_decorator
    def generate_images(self, text, *, clip=None, visual=None, mask=None, filter_thres=0.5, temperature=1.,
                        erase_visual=False, vc_mode=None, face_mode=None, use_cache=True, **kwargs):
        """dalle_artv.py:236-304.  use_cache=True (default): the prompt is run once and each sampled token then costs
        one incremental step over the per-layer key/value cache; use_cache=False: the reference's algorithm, the
        whole transformer over the growing prefix for every token.  Same sampling distribution either way (the
        two differ in bf16 summation order only; tests/test_models_gpu.py compares the logits step by step)."""
        tsl, total_len = self.text_seq_len, self.text_seq_len + self.target_seq_len
        text = text[:, :tsl]
        out = text
        # the visual control tokens do not change during sampling: tokenise once (the reference re-encodes them
        # every step, dalle_artv.py:464-466, with identical results)
        vis_tok = None
        if exists(visual) and not is_empty(visual):
            vis_tok = self.get_image_tokens(visual, which_vae='cvae')
            if erase_visual:
                vis_tok = self.random_erase_codebook(vis_tok, self.eraser, True)
        if use_cache:
            B = text.shape[0]
            cache = self.transformer.new_kv_cache(B, self.total_seq_len, text.device)
            # prompt = <bos> text, visual: the same ids _hidden builds (dalle_artv.py:441-477)
            text_range = torch.arange(tsl, device=text.device) + (self.num_text_tokens - tsl)
            tx = F.pad(torch.where(text == 0, text_range, text), (1, 0), value=0)
            vz = vis_tok if vis_tok is not None else -torch.ones(B, self.visual_seq_len, device=text.device).long()
            visual_range = torch.arange(self.visual_seq_len, device=text.device) + (self.num_visual_tokens - self.visual_seq_len)
            vz = torch.where(vz == -1, visual_range, vz)
            prompt = torch.cat((tx, vz), 1)
            h = self.transformer.prefill(self._embed_rows(prompt, 0), cache)[:, -1, :]
            if self.stable:
                h = self.norm_by_max(h)
            pos = prompt.shape[1]  # index of the position being sampled for
            # everything a step needs, prepared once: positional rows, the image block of the logits matrix, the session
            pos_rows = torch.cat([self.text_pos_emb.weight, self.visual_pos_emb.table(), self.image_pos_emb.table()], 0)
            c0, c1 = self._allowed_range(pos)
            lin, ln = self.to_logits[1], self.to_logits[0]
            w_img, b_img = self._w16()[c0:c1].contiguous(), lin.bias[c0:c1].contiguous()
            k_keep = max(int((1 - filter_thres) * lin.weight.shape[0]), 1)  # top_k() keeps this many of ALL logits
            sess = self.transformer.decode_session(cache, pos)
            toks = []
            for step in range(self.target_seq_len):
                # logits of the allowed (image) block only: the other columns are masked to -max by the reference and end
                # up with probability exactly 0 (dalle_artv.py:285-290), whatever top-k does, as long as k covers the block
                hn, _, _ = ops.layernorm_fwd(h.contiguous(), ln.weight, ln.bias, 1e-5, save_stats=False)
                blk = ops.gemm(hn, w_img, bias=b_img, out_dtype=torch.float32)
                if k_keep < c1 - c0:
                    blk = top_k(blk, thres=1.0 - k_keep / (c1 - c0))
                probs = torch.zeros(B, lin.weight.shape[0], device=text.device)
                probs[:, c0:c1] = F.softmax(blk / temperature, dim=-1)
                sample = torch.multinomial(probs, 1) - self.num_control_tokens  # same call, same RNG use as the reference
                toks.append(sample)
                if step == self.target_seq_len - 1:
                    break
                x_new = self.image_emb.weight[sample[:, 0]] + pos_rows[pos]
                h = sess.step(x_new)
                if self.stable:
                    h = self.norm_by_max(h)
                pos += 1
            out = torch.cat([out] + toks, dim=-1)
        else:
            for cur_len in range(out.shape[1], total_len):
                image = out[:, tsl:]
                hidden, _, _, _ = self._hidden(out[:, :tsl], vis_tok, image, False, False, None, None, None)
                sample = self._sample(hidden[:, -1, :], hidden.shape[1] - 1, filter_thres, temperature)
                out = torch.cat((out, sample), dim=-1)
        img_seq = out[:, -self.target_seq_len:].reshape(-1, self.image_seq_len)
        images = self.vae.decode(img_seq)
        if self.num_targets > 1:
            images = images.view(-1, self.num_targets, *images.shape[1:])
        if exists(clip):
            return images, clip(out[:, :tsl], images, return_loss=False)
        return images, [], None
