"""Host-side mirror of mmvid_pytorch/dalle_bert.py::BERT (259-1127) over the HIP kernels.

Same constructor, state_dict keys, attributes and methods as the reference class, so an MMVID-style
train.py / test.py drives it unchanged:
    forward(text, visual=, target=, return_loss=, rel=, vid=, ...) -> control_emb | (loss_msm, loss_rel, loss_vid)
    generate_images(text, visual=, mask_predict_steps=, mp_config=, ...) -> (images, pnag_samples, img_seq)
    get_image_tokens / recon_images / get_codebook_emb / decode_images / decode_masks / transformer_forward
What differs is where everything runs.  A training forward is a short chain of device launches:

    frontend (csrc/frontend.hip)   MSM masks + the VID negative's pixel warp, drawn on the device (992-1029, 204-238)
    VQGAN encode                   target and warped frames as ONE batch of 2*B*T frames (983, 1095)
    bert_build_ids (sample.hip)    every token id of the MSM / REL-negative / VID-negative sequences + the CE rows
    assemble + tower               the three passes (1037, 1061, 1101) as one batch of 3B sequences
    BertHeads                      to_logits + CE, to_logits_rel / _vid + BCE, forward and backward

and sampling (`generate_images` -> mmvid_amd/sampling.py) runs every video and beam candidate of a mask-predict step
as one tower batch with all state on the device.  Tests inject the stochastic choices through the private
`_mask1` / `_target_warp` kwargs.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops, sampling
from .clip_tower import OpenAICLIPTransformer
from .frontend import Frontend, face_choices
from .functional import PosTable, AssembleSequence, BertHeads, LNLinear, Linear, LayerNormRows
from .modules import AxialPositionalEmbedding, AxialPositionalEmbeddingList


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


class BERT(nn.Module):
    TEXT_ID_LOG_MAX = 16  # forwards between two zero_grad() calls whose text ids are kept for the row-wise gradient exchange

    def __init__(self, *, dim, vae, cvae=None, num_text_tokens=10000, text_seq_len=256, stable=False,
                 text_feature_dim=0, fixed_language_model=None, which_transformer='none', num_visuals=1,
                 num_targets=1, use_separate_visual_emb=False, insert_sep=False, text_emb_bottleneck=False,
                 **kwargs):
        super().__init__()
        image_size = vae.image_size
        num_image_tokens = vae.num_tokens
        image_fmap_size = vae.image_size // (2**vae.num_layers)
        image_seq_len = image_fmap_size**2
        self.dim, self.num_visuals, self.num_targets = dim, num_visuals, num_targets
        # torchvision RandomErasing parameters of the reference (dalle_bert.py:290-294, 427-432), consumed by the
        # device front-end: (p, scale, ratio)
        self.random_erasing = dict(p=1.0, scale=(0.2, 0.8), ratio=(0.5, 2.0))
        self.visual_eraser = dict(p=0.95, scale=(0.55, 0.85), ratio=(0.5, 2.0))

        if fixed_language_model is None:
            num_text_tokens = num_text_tokens + text_seq_len  # unique pad id per position (dalle_bert.py:299)
            self.text_emb = nn.Embedding(num_text_tokens, dim)
            self.text_pos_emb = nn.Embedding(text_seq_len, dim)
            self.text_feature_mapping = None
        else:
            # dalle_bert.py:307-322: the text is ONE token, a sentence feature of a frozen language model (computed by the driver,
            # utils_train.py:194-215) mapped to `dim`; there is no text table and no text position.
            assert text_feature_dim > 0
            text_seq_len, num_text_tokens = 1, 1
            self.text_emb = self.text_pos_emb = None
            if text_emb_bottleneck is not None:  # (sic) the class default False reaches int(False) = 0 in the reference
                nf = int(text_emb_bottleneck)
                if nf <= 0 or nf % 8 or text_feature_dim % 8 or max(nf, text_feature_dim, dim) > 1024:
                    raise ValueError(f'text_emb_bottleneck={text_emb_bottleneck!r}, text_feature_dim={text_feature_dim}: the mapping '
                                     'kernels take widths that are multiples of 8, at most 1024 (pass text_emb_bottleneck=None '
                                     'for the single Linear)')
                self.text_feature_mapping = nn.Sequential(nn.LayerNorm(text_feature_dim), nn.Linear(text_feature_dim, nf),
                                                          nn.LayerNorm(nf), nn.Linear(nf, dim), nn.LayerNorm(dim))
            else:
                if text_feature_dim % 8:
                    raise ValueError(f'text_feature_dim={text_feature_dim} must be a multiple of 8')
                self.text_feature_mapping = nn.Linear(text_feature_dim, dim)
            # what the text segment reads from the (absent) table and position: a zero row, the mapped feature is added on top
            self.register_buffer('_no_text_row', torch.zeros(1, dim), persistent=False)
        self.text_feature_dim = text_feature_dim
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
        self.fixed_language_model = fixed_language_model
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
        self.frontend = Frontend(seed=kwargs.get('frontend_seed'))  # None: torch.initial_seed() + rank at first use (train.py:87)
        self._w16_cache = {}
        self._row_cache = {}
        self._debug_keep = None
        self._text_id_log, self._text_id_overflow = [], False
        # segment table: which embedding table each position reads (0 special, 1 text, 2 visual, 3 image)
        seg = [0] + [1] * self.text_seq_len + [2] * self.visual_seq_len + [0, 0] + [3] * self.target_seq_len
        self.register_buffer('_seg', torch.tensor(seg, dtype=torch.int32), persistent=False)

    def half(self):
        """train.py:194-195 (`--fp16`).  The kernels always compute in bf16 on the MFMA pipe over fp32 master weights
        (what mixed precision buys is already in place); fp16 parameters would have no kernel to run on."""
        import warnings
        warnings.warn('mmvid_amd: .half() ignored -- compute is bf16 MFMA over fp32 master weights (fp16 checkpoints still '
                      'load: values are widened on copy)', UserWarning)
        return self

    # ------------------------------------------------------------------------------------ small helpers
    def _w16(self, lin):
        """bf16 copy of a head's Linear weight: the view a fused optimiser keeps current (attach_head_shadow), else a
        cast refreshed whenever the parameter's storage or version changes."""
        w = lin.weight
        c = self._w16_cache.get(id(lin))
        if c is not None and c[0] == 'attached':
            if c[2] == w.data_ptr():
                if c[3] != w._version:  # written through torch (load_state_dict, an in-place op): refresh the view itself
                    ops.cast_bf16(w.detach().contiguous(), c[1])
                    self._w16_cache[id(lin)] = ('attached', c[1], c[2], w._version)
                return c[1]
            c = None  # the parameter moved (.to(), load_state_dict(assign=True)): the attachment is void
        key = (w._version, w.data_ptr())
        if c is None or c[0] != key:
            c = (key, ops.cast_bf16(w.detach().contiguous()))
            self._w16_cache[id(lin)] = c
        return c[1]

    def attach_head_shadow(self, lin, view):
        """Engine hook: `view` is kept equal to bf16(lin.weight) by the fused optimiser for as long as the parameter
        stays in the trainer's flat buffer (same data_ptr)."""
        self._w16_cache[id(lin)] = ('attached', view, lin.weight.data_ptr(), lin.weight._version)

    def head_shadow_targets(self):
        """Linear layers whose weight the MFMA kernels read in bf16 (engine.FlatTrainer attaches shadow views)."""
        m = self.text_feature_mapping
        extra = [] if m is None else ([m] if isinstance(m, nn.Linear) else [m[1], m[3]])
        return [self.to_logits[1]] + extra

    def _tables(self):
        vis = self.visual_emb.weight if (self.num_visuals > 0 and self.visual_emb is not None) else self.image_emb.weight
        text = self.text_emb.weight if self.text_emb is not None else self._no_text_row
        return (self.special_emb.weight, text, vis, self.image_emb.weight)

    def sparse_grad_rows(self):
        """Tables whose gradient has few non-zero rows per step, with the row ids of the last forward (engine.FlatTrainer
        exchanges them row-wise instead of all-reducing 152 MB of mostly zeros)."""
        if self.text_emb is None:
            return {}
        log = self._text_id_log
        if not log or self._text_id_overflow:  # nothing logged / more forwards since zero_grad than the log holds: the
            return {'text_emb.weight': None}   # trainer falls back to the dense all-reduce for this table
        return {'text_emb.weight': log[0] if len(log) == 1 else torch.cat(log)}

    def _note_table_backward(self, _g):
        self.table_grad_pending = True

    # public: True from the moment a backward of this model has (possibly) written rows of a table gradient until the trainer has
    # accounted for them (FlatTrainer.step() records them as dirty, zero_grad() clears them).  A backward whose forward was logged
    # BEFORE the last zero_grad() is caught by this flag and by nothing else (forward, zero_grad, backward, zero_grad).
    table_grad_pending = False

    def reset_sparse_grad_rows(self):
        """Called by FlatTrainer.zero_grad(): the gradients start from zero, so does the list of touched rows."""
        self._text_id_log, self._text_id_overflow = [], False

    def _log_text_ids(self, ids):
        """Every text-segment id of a grad-enabled forward: the only rows of text_emb its backward can touch.  Logged per
        forward since the last zero_grad so gradient accumulation stays covered; past TEXT_ID_LOG_MAX forwards the log is
        declared incomplete (sparse_grad_rows -> None -> dense all-reduce) rather than silently dropping the oldest."""
        if not torch.is_grad_enabled() or self.text_emb is None:
            return
        if len(self._text_id_log) >= self.TEXT_ID_LOG_MAX:
            self._text_id_overflow = True
            return
        self._text_id_log.append(ids[:, 1:1 + self.text_seq_len].reshape(-1))

    def _pos_layout(self):
        """Segments of the positional table for functional.PosTable: (dst0, rows, src0, params, axial dims)."""
        sp, f = self.special_pos_emb.weight, self.image_fmap_size
        text_pos = self.text_pos_emb.weight if self.text_pos_emb is not None else self._no_text_row
        lay = [(0, 1, 0, (sp, ), ()), (1, self.text_seq_len, 0, (text_pos, ), ())]
        at = 1 + self.text_seq_len
        if self.num_visuals > 0:
            for m in self.visual_pos_emb.module_list:  # one (h, w) axial table per visual frame, + a zero [SEP] row each
                lay.append((at, f * f, 0, (m.weights_0, m.weights_1), (f, f)))  # [1,f,1,E] / [1,1,f,E]: rows of E in memory
                at += f * f + (1 if self.insert_sep else 0)
        lay.append((at, 2, 1, (sp, ), ()))
        at += 2
        tp, T = self.target_pos_emb, self.num_targets
        lay.append((at, T * f * f, 0, (tp.weights_0, tp.weights_1, tp.weights_2), (T, f, f)))
        return lay, at + T * f * f

    def _pos_table(self):
        """[total_seq_len, dim]: special_pos / text_pos / axial tables laid out along the sequence."""
        sp = self.special_pos_emb.weight
        if sp.is_cuda:  # one launch forward, one backward (functional.PosTable); the torch construction below is the host form
            lay, L = self._pos_layout()
            params = [w for seg in lay for w in seg[3]]
            return PosTable.apply(lay, L, *params)
        parts = [sp[0:1], (self.text_pos_emb.weight if self.text_pos_emb is not None else self._no_text_row)[:self.text_seq_len]]
        if self.num_visuals > 0:
            parts.append(self.visual_pos_emb.table(insert_sep=bool(self.insert_sep)))
        parts += [sp[1:3], self.target_pos_emb.table()]
        return torch.cat(parts, 0)

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

    def swap_one_frame_along_batch(self, tokens, t=1):
        """dalle_bert.py:854-866 (a helper no forward path calls): in every sample one randomly chosen frame of the [b, n, c]
        token embeddings is replaced by the frame picked in the sample half a batch away.  The frame index is drawn on
        the device (the reference uses numpy's global generator)."""
        b, n, c = tokens.shape
        out = tokens.detach().clone().reshape(b, t, n // t, c)
        rows = torch.arange(b, device=tokens.device)
        idx = torch.randint(0, t, (b, ), device=tokens.device)
        picked = out[rows, idx]
        h = (b + 1) // 2  # torch.chunk(x, 2)[::-1]: the second (shorter) half first
        out[rows, idx] = torch.cat((picked[h:], picked[:h]), 0)
        return out.reshape(b, n, c)

    def decode_images(self, img_seq):
        return self.vae.decode(img_seq.reshape(-1, self.image_seq_len))

    def decode_masks(self, mask):
        f = self.image_fmap_size
        patch = self.image_size // f
        up = mask.reshape(-1, 1, f, f).repeat_interleave(patch, 2).repeat_interleave(patch, 3)
        return F.pad(up, (0, 0, 0, 0, 0, 2))  # one channel of "red" + two empty ones

    def random_erase_codebook(self, image, eraser, erase_half=False):
        """dalle_bert.py:779-794 on the device: one RandomErasing box per sample (same box on every frame) set to [MASK],
        or the lower half of every frame."""
        f = self.image_fmap_size
        image = image.contiguous()
        tv = image.shape[1] // (f * f)
        return self.frontend.random_erase(image, tv, f, self.image_token_lut['[MASK]'], eraser['p'], eraser['scale'],
                                          eraser['ratio'], erase_half)

    def erase_codebook_face(self, image, vc_mode, face_mode=None):
        """dalle_bert.py:796-848 on the device (region tables in frontend.face_choices)."""
        f = self.image_fmap_size
        image = image.contiguous()
        tv = image.shape[1] // (f * f)
        choices, frame0 = face_choices(vc_mode, face_mode)
        return self.frontend.erase_choice(image, tv, f, self.image_token_lut['[MASK]'], choices, frame0)

    # ------------------------------------------------------------------------------ sequence assembly
    def _visual_tokens(self, visual, erase_visual, erase_visual_half, vc_mode, face_mode, visual_aug_mode):
        """Token ids of the visual control segment (dalle_bert.py:933-957), or None = all [MASK]."""
        if self.num_visuals == 0 or not (exists(visual) and len(visual)):
            return None
        if visual_aug_mode == 'motion_color' and torch.is_tensor(visual) and visual.dim() == 5:
            # scripts/mmvoxceleb/image_and_video/train.sh:10; dalle_bert.py:940-943 / dalle_artv.py:460-463: colour jitter of the
            # video part (frames 1..) of the visual control, gated at 0.9 per call -- drawn on the device
            visual = self.frontend.visual_color_jitter(visual, 0.9, 1)
        tok = self.get_image_tokens(visual, insert_sep=self.insert_sep, which_vae='cvae')
        if erase_visual:
            tok = self.random_erase_codebook(tok, self.visual_eraser, erase_visual_half)
        if vc_mode is not None:
            tok = self.erase_codebook_face(tok, vc_mode, face_mode)
        return tok

    def _assemble(self, ids, length, text_rows=None):
        pos = self._pos_table()[:length]
        x = AssembleSequence.apply(pos, ids.contiguous(), self._seg[:length].contiguous(), *self._tables())
        if x.requires_grad:  # the backward of this node scatter-adds into the tables' gradients: say so when it runs (see below)
            x.register_hook(self._note_table_backward)
        if text_rows is not None:  # fixed language model: the text position holds the mapped feature (dalle_bert.py:924-925)
            x[:, self.txt_tok_index].add_(text_rows)
        return x

    def _map_text_feature(self, feat):
        """text_feature_mapping (dalle_bert.py:312-322) on [B, text_feature_dim] fp32 rows -> [B, dim]."""
        m = self.text_feature_mapping
        if isinstance(m, nn.Linear):
            return Linear.apply(feat, m.weight, m.bias, self._w16(m))
        h = LNLinear.apply(feat, m[0].weight, m[0].bias, m[1].weight, m[1].bias, self._w16(m[1]))
        h = LNLinear.apply(h, m[2].weight, m[2].bias, m[3].weight, m[3].bias, self._w16(m[3]))
        return LayerNormRows.apply(h, m[4].weight, m[4].bias)

    def _head_rows(self, B, nseq, device):
        """Row numbers (into the [nseq*B*L, dim] tower output) and labels of the REL / VID heads: positives from the MSM
        pass, negatives from their own pass.  Built once per shape (device constants: the step stays capturable)."""
        key = (B, nseq, str(device))
        if key not in self._row_cache:
            L = self.total_seq_len
            b = torch.arange(B, device=device)
            lab = torch.cat((torch.ones(B, device=device), torch.zeros(B, device=device)))
            rows = {}
            for which, at in (('rel', self.rel_tok_index), ('vid', self.vid_tok_index)):
                rows[which] = [torch.cat((b * L + at, (s * B + b) * L + at)).contiguous() for s in range(nseq)]
            self._row_cache[key] = (rows, lab)
        return self._row_cache[key]

    # ----------------------------------------------------------------------------------------- forward
    def forward(self, text, visual=None, target=None, mask=None, return_loss=False, rel=False, vid=False,
                erase_visual=False, erase_visual_half=False, msm_strategy_prob=[0.7, 0.1, 0.1, 0.1],
                msm_bernoulli_prob=[0.2, 0.5], rel_no_fully_masked=False,
                vid_strategy_prob=[0.25, 0.25, 0.25, 0.25], negvc=False, visual_neg=None, text_neg=None, pc_prob=0,
                vc_mode=None, face_mode=None, visual_aug_mode=None, _mask1=None, _target_warp=None, **kwargs):
        device = text.device
        B = text.shape[0]
        text_rows = None
        if self.fixed_language_model is None:
            assert text.shape[-1] == self.text_seq_len, \
                f'the length {text.shape[-1]} of the text tokens you passed in does not have the correct length ({self.text_seq_len})'
            text = ops._chk(text.contiguous(), torch.int64, 'text')
        else:  # `text` is the sentence feature [B, text_feature_dim] (dalle_bert.py:897-898, 924-925); its one token reads row 0
            if negvc:
                raise NotImplementedError('negvc with a fixed language model: the reference indexes a text table it does not '
                                          'have in this mode (dalle_bert.py:930-932)')
            assert text.dim() == 2 and text.shape[-1] == self.text_feature_dim, \
                f'expected text features of shape [B, {self.text_feature_dim}], got {tuple(text.shape)}'
            text_rows = self._map_text_feature(ops._chk(text.contiguous().float(), torch.float32, 'text'))
            text = torch.zeros(B, 1, dtype=torch.int64, device=device)
        MASK = self.image_token_lut['[MASK]']
        pad_base = self.num_text_tokens - self.text_seq_len
        vis_tok = self._visual_tokens(visual, erase_visual, erase_visual_half, vc_mode, face_mode, visual_aug_mode)
        if not return_loss:  # control embedding only (dalle_bert.py:977-978)
            empty = torch.empty(B, 0, dtype=torch.long, device=device)
            ids = ops.bert_build_ids(text, vis_tok, self.visual_seq_len, empty, None, torch.empty(B, 0, dtype=torch.uint8, device=device),
                                     pad_base, MASK, False, False)[0]
            self.frontend.advance(device)
            self._log_text_ids(ids)
            return self._assemble(ids, self.control_seq_len, text_rows)

        T, f = self.num_targets, self.image_fmap_size
        do_vid = vid and T > 1
        # ---- stochastic choices: on the device, or injected by tests
        if _mask1 is None:
            mask1, not_fully_masked = self.frontend.msm_masks(B, T, f, device, msm_strategy_prob, msm_bernoulli_prob, pc_prob)
        else:
            mask1 = _mask1.to(device=device, dtype=torch.uint8).contiguous()
            not_fully_masked = kwargs.get('_not_fully_masked', torch.ones(B, device=device))
        # ---- tokens: the target and its warped negative go through the VQGAN as one batch (983, 1095)
        target_warp = None
        if do_vid and torch.is_tensor(target) and target.dim() == 5 and _target_warp is None:
            # The negative is the target with ONE frame's pixels changed (or its frames permuted), and the VQGAN tokenises
            # frames independently: encode the B*T target frames plus the B new frames as one batch, then assemble the
            # negative's tokens from them.  Bit-identical to tokenising the warped video (984-985 + 1094-1095), 44 % fewer
            # encoder frames (tests/test_parity_gpu.py::test_vid_negative_tokens_without_reencoding).
            frames = ops._chk(target.contiguous().float(), torch.float32, 'target')
            _, _, C, H, W = frames.shape
            both = torch.empty(B * T + B, C, H, W, device=device, dtype=torch.float32)
            both[:B * T].copy_(frames.view(B * T, C, H, W))
            self.frontend.vid_warp_new_frames(frames, vid_strategy_prob, both[B * T:])
            toks = self.vae.get_codebook_indices(both)
            target = toks[:B * T].reshape(B, -1).contiguous()
            target_warp = self.frontend.vid_warp_tokens(target, toks[B * T:], T)
        elif do_vid and torch.is_tensor(target) and target.dim() == 5:
            both = torch.empty((2 * B, ) + tuple(target.shape[1:]), device=device, dtype=torch.float32)
            both[:B].copy_(target)
            both[B:].copy_(_target_warp)
            toks = self.get_image_tokens(both)
            target, target_warp = toks[:B].contiguous(), toks[B:].contiguous()
        else:
            target = self.get_image_tokens(target).contiguous()
            if do_vid:
                tw = _target_warp if _target_warp is not None else target  # token-level targets: nothing to warp
                target_warp = self.get_image_tokens(tw.to(device)).contiguous()
        self.frontend.advance(device)

        # ---- every id of the 1-3 sequences + the CE rows, one launch; then one tower pass over all of them
        if rel:
            assert B >= 2 and B % 2 == 0  # for REL swapping (dalle_bert.py:1045-1046)
        text_neg_ids = None
        if rel and negvc:
            # `visual_neg` is accepted and ignored, as in the reference (dalle_bert.py:869-892 takes it, nothing reads it)
            text_neg_ids = ops._chk(text_neg.contiguous(), torch.int64, 'text_neg')
        ids, sel, tfull, cnt = ops.bert_build_ids(text, vis_tok, self.visual_seq_len, target, target_warp, mask1, pad_base, MASK,
                                                  bool(rel), bool(do_vid), text_neg=text_neg_ids)
        self._log_text_ids(ids)  # rows of text_emb this forward's backward can touch (sparse_grad_rows)
        if text_rows is not None and (rel or do_vid):  # one row per sequence of the batched pass: MSM, REL (halves swapped), VID
            half = B // 2
            per_pass = [text_rows] + ([torch.cat((text_rows[half:], text_rows[:half]))] if rel else []) + \
                ([text_rows] if do_vid else [])
            text_rows = torch.cat(per_pass)
        x_seq = self._assemble(ids, self.total_seq_len, text_rows)
        if text_neg_ids is not None and self.num_visuals > 0 and self.visual_seq_len > 0:
            # negvc with a visual control (dalle_bert.py:908-909, 927-935, 974-975, 1047-1054): the reference's control_neg is
            # [REL] + text_neg + [ST1] [VID] WITHOUT the visual segment, so its REL-negative pass is a shorter sequence (the tower's
            # restricted rows keep their absolute positions: clip_model.py:218-222 slices the mask).  The REL third of the batch is
            # assembled at full length like the others, its visual positions are dropped, and it goes through the tower on its own;
            # the result returns to its full-length rows (the dropped positions stay zero: the heads read row 0 of that pass only).
            L = self.total_seq_len
            v0 = 1 + self.text_seq_len
            keep = torch.cat((torch.arange(0, v0, device=device), torch.arange(v0 + self.visual_seq_len, L, device=device)))
            x_neg = x_seq[B:2 * B].index_select(1, keep).contiguous()
            x_main = torch.cat((x_seq[:B], x_seq[2 * B:])) if do_vid else x_seq[:B]
            y_main = self.transformer_forward(x_main.contiguous())
            y_neg = self.transformer_forward(x_neg)
            y_rel = torch.zeros(B, L, y_neg.shape[-1], device=device, dtype=y_neg.dtype).index_copy(1, keep, y_neg)
            y = torch.cat((y_main[:B], y_rel, y_main[B:])) if do_vid else torch.cat((y_main, y_rel))
        else:
            y = self.transformer_forward(x_seq)  # [nseq*B, L, dim]
        if self._debug_keep is not None:  # tools/stress_nan2.py: the stage tensors of the last (replayed) forward
            self._debug_keep.update(mask1=mask1, nfm=not_fully_masked, target=target, target_warp=target_warp, ids=ids, sel=sel,
                                    tfull=tfull, cnt=cnt, x_seq=x_seq, y=y)
        nseq = 1 + int(bool(rel)) + int(bool(do_vid))
        rows, labels = self._head_rows(B, nseq, device)
        lin = self.to_logits[1]
        loss_msm, loss_rel, loss_vid, logits = BertHeads.apply(
            y, tfull, sel, cnt, not_fully_masked, labels, rows['rel'][1] if rel else None,
            rows['vid'][nseq - 1] if do_vid else None, bool(rel_no_fully_masked), B, self._w16(lin),
            self.to_logits[0].weight, self.to_logits[0].bias, lin.weight, lin.bias,
            self.to_logits_rel[0].weight, self.to_logits_rel[0].bias, self.to_logits_rel[1].weight, self.to_logits_rel[1].bias,
            self.to_logits_vid[0].weight, self.to_logits_vid[0].bias, self.to_logits_vid[1].weight, self.to_logits_vid[1].bias)
        self._last_logits_msm = logits.view(B, self.total_seq_len, -1)[:, self.control_seq_len:]
        return loss_msm, loss_rel, loss_vid

    # ---------------------------------------------------------------------------------------- sampling
    @torch.no_grad()
    @eval_decorator
    def generate_images(self, text, *, visual=None, mask=None, img=None, argmax=False, dynamic=True, debug=False,
                        erase_visual=False, mask_predict_steps=10, preserve=None, t_overlap=1, pc_mode=None,
                        vc_mode=None, face_mode=None, mp_config=None, long_mode='long', **kwargs):
        """dalle_bert.py:434-487 -> (images [b,T,3,H,W], pnag_samples, img_seq [(b T), n])."""
        control_emb = self(text, visual=visual, erase_visual=erase_visual, erase_visual_half=True, vc_mode=vc_mode,
                           face_mode=face_mode, return_loss=False)
        img_seq, pnag_samples = self.mask_predict(control_emb, argmax=argmax, dynamic=dynamic, debug=debug,
                                                  steps=mask_predict_steps, preserve=preserve, t_overlap=t_overlap,
                                                  pc_mode=pc_mode, mp_config=mp_config, long_mode=long_mode, **kwargs)
        img_seq = img_seq.reshape(-1, self.image_seq_len)
        images = self.vae.decode(img_seq)
        return images.view(-1, self.num_targets, *images.shape[1:]), pnag_samples, img_seq

    @torch.no_grad()
    def mask_predict(self, control_emb, dynamic=True, debug=False, steps=10, preserve=None, t_overlap=1,
                     mp_config=None, long_mode='long', _race=None, _trace=None, **kwargs):
        """dalle_bert.py:514-714, batched over videos and beam candidates on the device (mmvid_amd/sampling.py)."""
        return sampling.mask_predict(self, control_emb, dynamic=dynamic, debug=debug, steps=steps, preserve=preserve,
                                     t_overlap=t_overlap, mp_config=mp_config, long_mode=long_mode, race=_race, trace=_trace)
