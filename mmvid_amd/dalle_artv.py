"""Host-side mirror of mmvid_pytorch/dalle_artv.py::DALLE (103-542), the autoregressive "ART-V" baseline, over
the HIP kernels: same constructor, state_dict keys, `forward` (logits or (loss, 0, 0)) and `generate_images`
-> (images, [], None).

The reference's block-diagonal vocabulary mask (a [1, L, 51584] bool buffer, dalle_artv.py:215-227, applied with
masked_fill at 509-512) says: text positions predict text ids, visual positions visual ids, image positions image
ids; every other class gets logit -max, i.e. probability exactly 0.  Here that structure is used instead of
materialised: each of the three position segments runs `to_logits` against ITS block of the weight only (LayerNorm +
MFMA GEMM + the fused cross-entropy kernels), which is the same loss with ~14x fewer head FLOPs and no [B, L, 51584]
tensors.  Sampling keeps a per-layer key/value cache (the reference recomputes the whole prefix per token) and draws
tokens with the device sampler of csrc/sample.hip."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, ops
from .clip_tower import OpenAICLIPTransformer
from .dalle_bert import DivideMax, eval_decorator, exists, set_requires_grad
from .frontend import Frontend, face_choices
from .functional import AssembleSequence, LNLinear, LNLinearCrossEntropy
from .modules import AxialPositionalEmbedding, AxialPositionalEmbeddingList

NEG = -torch.finfo(torch.float32).max


def is_empty(t):
    return t.nelement() == 0


def top_k(logits, thres=0.5):
    """dalle_artv.py:61-67: keep the k = max(int((1 - thres) * n), 1) largest logits of each row, -inf elsewhere."""
    k = max(int((1 - thres) * logits.shape[-1]), 1)
    kept = torch.full_like(logits, float('-inf'))
    val, ind = torch.topk(logits, k)
    return kept.scatter_(1, ind, val)


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
        self.loss_vis_weight, self.loss_img_weight = 1., loss_img_weight
        self.eraser = dict(p=1.0, scale=(0.4, 0.8), ratio=(0.5, 2.0))  # RandomErasing(value=-1), dalle_artv.py:229-232
        self.frontend = Frontend(seed=kwargs.get('frontend_seed'))
        self._w16_cache = None
        seg = [0] * (text_seq_len + 1) + [1] * self.visual_seq_len + [2] * self.target_seq_len
        self.register_buffer('_seg', torch.tensor(seg, dtype=torch.int32), persistent=False)

    def half(self):
        """train.py:194-195 (`--fp16`).  The kernels always compute in bf16 on the MFMA pipe over fp32 master weights
        (what mixed precision buys is already in place); fp16 parameters would have no kernel to run on."""
        import warnings
        warnings.warn('mmvid_amd: .half() ignored -- compute is bf16 MFMA over fp32 master weights (fp16 checkpoints still '
                      'load: values are widened on copy)', UserWarning)
        return self

    # ---- the vocabulary blocks --------------------------------------------------------------------------------------
    @property
    def logits_mask(self):
        """The reference's [1, total_seq_len, total_tokens] bool buffer, materialised on demand (tests only)."""
        m = torch.block_diag(torch.ones(self.text_seq_len, self.num_text_tokens),
                             torch.ones(self.visual_seq_len, self.num_visual_tokens),
                             torch.ones(self.target_seq_len, self.num_image_tokens)) == 0
        return m.unsqueeze(0)

    def _segments(self, L):
        """[(first position, end position, first class, end class)] of the block-diagonal mask, clipped to L positions."""
        tl, cl = self.text_seq_len, self.control_seq_len
        segs = [(0, tl, 0, self.num_text_tokens), (tl, cl, self.num_text_tokens, self.num_control_tokens),
                (cl, self.total_seq_len, self.num_control_tokens, self.total_tokens)]
        return [(lo, min(hi, L), c0, c1) for lo, hi, c0, c1 in segs if lo < L]

    def _allowed_range(self, pos):
        """classes position `pos` may predict (dalle_artv.py:215-219)."""
        for lo, hi, c0, c1 in self._segments(self.total_seq_len):
            if lo <= pos < hi:
                return c0, c1
        raise IndexError(pos)

    # ---- bf16 copy of the 51,584 x 768 head ----------------------------------------------------------------------------
    def _w16(self):
        w = self.to_logits[1].weight
        c = self._w16_cache
        if c is not None and c[0] == 'attached':
            if c[2] == w.data_ptr():
                if c[3] != w._version:  # written through torch since the attachment: refresh the attached view
                    ops.cast_bf16(w.detach().contiguous(), c[1])
                    self._w16_cache = ('attached', c[1], c[2], w._version)
                return c[1]
            c = None
        key = (w._version, w.data_ptr())
        if c is None or c[0] != key:
            c = self._w16_cache = (key, ops.cast_bf16(w.detach().contiguous()))
        return c[1]

    def attach_head_shadow(self, lin, view):
        """Engine hook (see BERT.attach_head_shadow): a fused optimiser updates parameters through raw pointers, which
        never bumps `_version`; it keeps `view` == bf16(weight) itself."""
        assert lin is self.to_logits[1]
        self._w16_cache = ('attached', view, lin.weight.data_ptr(), lin.weight._version)

    def head_shadow_targets(self):
        return [self.to_logits[1]]

    # ---- token helpers (dalle_artv.py:306-416) --------------------------------------------------------------------------
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

    @torch.no_grad()
    def recon_images(self, images, which_vae='vae'):
        """dalle_artv.py:344-354: frames -> tokens -> frames through the (control) VQGAN."""
        vae = self.cvae if (which_vae == 'cvae' and self.cvae is not None) else self.vae
        return vae.decode(self.get_image_tokens(images, reshape=False, which_vae=which_vae))

    def random_erase_codebook(self, image, eraser, erase_half=False):
        f = self.image_fmap_size
        image = image.contiguous()
        return self.frontend.random_erase(image, image.shape[1] // (f * f), f, -1, eraser['p'], eraser['scale'],
                                          eraser['ratio'], erase_half)

    def erase_codebook_face(self, image, vc_mode, face_mode=None):
        """dalle_artv.py:356-416; erased positions become -1 (later replaced by per-position pad ids, 473-477)."""
        f = self.image_fmap_size
        image = image.contiguous()
        if vc_mode == 'face3_8x8':
            raise NotImplementedError(vc_mode)  # BERT-only mode
        choices, frame0 = face_choices(vc_mode, face_mode)
        if vc_mode in ('mask_8x8', 'mask2_8x8') and face_mode is None:
            # the reference builds strategy 2's masked copy but never assigns it (dalle_artv.py:401-404): unchanged
            choices = [(choices[0][0] + choices[1][0], 0, (0, 0, 0, 0)), choices[2]]
        return self.frontend.erase_choice(image, image.shape[1] // (f * f), f, -1, choices, frame0)

    def _visual_tokens(self, visual, erase_visual, erase_visual_half, vc_mode, face_mode, visual_aug_mode):
        if not (exists(visual) and not is_empty(visual)):
            return None
        if visual_aug_mode == 'motion_color' and torch.is_tensor(visual) and visual.dim() == 5:
            # scripts/mmvoxceleb/image_and_video/train.sh:10; dalle_bert.py:940-943 / dalle_artv.py:460-463: colour jitter of the
            # video part (frames 1..) of the visual control, gated at 0.9 per call -- drawn on the device
            visual = self.frontend.visual_color_jitter(visual, 0.9, 1)
        tok = self.get_image_tokens(visual, which_vae='cvae')
        if erase_visual:
            tok = self.random_erase_codebook(tok, self.eraser, erase_visual_half)
        if vc_mode is not None:
            tok = self.erase_codebook_face(tok, vc_mode, face_mode)
        return tok

    # ---- sequence -> hidden states ---------------------------------------------------------------------------------------
    def _prompt_ids(self, text, vis_tok):
        """<bos> + text (0 -> per-position pad id) + visual tokens (-1 -> per-position pad id): dalle_artv.py:441-477."""
        device, B = text.device, text.shape[0]
        assert text.shape[-1] == self.text_seq_len, \
            f'the length {text.shape[-1]} of the text tokens you passed in does not have the correct length ({self.text_seq_len})'
        text_pad = torch.arange(self.text_seq_len, device=device) + (self.num_text_tokens - self.text_seq_len)
        text = F.pad(torch.where(text == 0, text_pad, text), (1, 0), value=0)
        if vis_tok is None:
            vis_tok = torch.full((B, self.visual_seq_len), -1, dtype=torch.long, device=device)
        vis_pad = torch.arange(self.visual_seq_len, device=device) + (self.num_visual_tokens - self.visual_seq_len)
        return text, torch.where(vis_tok == -1, vis_pad, vis_tok)

    def _pos_rows(self):
        return torch.cat([self.text_pos_emb.weight, self.visual_pos_emb.table(), self.image_pos_emb.table()], 0)

    def _hidden(self, text, vis_tok, image):
        text, visual = self._prompt_ids(text, vis_tok)
        parts = [text, visual]
        if exists(image) and not is_empty(image):
            image = self.get_image_tokens(image)
            parts.append(image)
        ids = torch.cat(parts, 1)
        if ids.shape[1] > self.total_seq_len:  # drop the last token when training (dalle_artv.py:496-498)
            ids = ids[:, :-1]
        L = ids.shape[1]
        x = AssembleSequence.apply(self._pos_rows()[:L], ids.contiguous(), self._seg[:L].contiguous(), self.text_emb.weight,
                                   self.visual_emb.weight, self.image_emb.weight)
        out = self.transformer(x)
        if self.stable:
            out = self.norm_by_max(out)
        return out, text, visual, image

    def _logits_rows(self, rows, cols=None):
        lin = self.to_logits[1]
        if cols is None:
            return LNLinear.apply(rows, self.to_logits[0].weight, self.to_logits[0].bias, lin.weight, lin.bias, self._w16())
        with torch.no_grad():  # one class block: inference only (training goes through LNLinearCrossEntropy)
            hn = ops.layernorm_fwd(rows.contiguous(), self.to_logits[0].weight, self.to_logits[0].bias, 1e-5, save_stats=False)[0]
            return ops.gemm(hn, self._w16()[cols[0]:cols[1]], bias=lin.bias.detach()[cols[0]:cols[1]], out_dtype=torch.float32)

    def forward(self, text, visual=None, target=None, return_loss=False, erase_visual=False, erase_visual_half=False,
                vc_mode=None, face_mode=None, visual_aug_mode=None, **kwargs):
        vis_tok = self._visual_tokens(visual, erase_visual, erase_visual_half, vc_mode, face_mode, visual_aug_mode)
        out, text, visual, image = self._hidden(text, vis_tok, target)
        B, L, E = out.shape
        if not return_loss:
            with torch.no_grad():  # the reference's dense [B, L, total_tokens] tensor, -max outside each position's block
                logits = torch.full((B, L, self.total_tokens), NEG, device=out.device, dtype=torch.float32)
                for lo, hi, c0, c1 in self._segments(L):
                    logits[:, lo:hi, c0:c1] = self._logits_rows(out[:, lo:hi].reshape(-1, E), (c0, c1)).view(B, hi - lo, c1 - c0)
            return logits
        assert exists(image), 'when training, image must be supplied'
        # labels (dalle_artv.py:519-524) relative to each segment's class block: text[1:], visual ids, image ids
        labels = (text[:, 1:], visual, image)
        lin, ln = self.to_logits[1], self.to_logits[0]
        losses = []
        for (lo, hi, c0, c1), lab in zip(self._segments(L), labels):
            rows = out[:, lo:hi].reshape(B * (hi - lo), E)
            loss, _ = LNLinearCrossEntropy.apply(rows, lab[:, :hi - lo].reshape(-1).contiguous(), None, ln.weight, ln.bias,
                                                 lin.weight, lin.bias, self._w16(), (c0, c1))
            losses.append(loss)
        loss = (losses[0] + self.loss_vis_weight * losses[1] + self.loss_img_weight * losses[2]) / \
            (self.loss_img_weight + self.loss_vis_weight + 1)
        zero = torch.tensor(0.0, device=text.device)
        return loss, zero, zero

    # ---- sampling ---------------------------------------------------------------------------------------------------------
    def _embed_rows(self, ids, first_pos):
        """Embedding + positional rows for `ids` [B, n] occupying positions first_pos .. first_pos+n-1."""
        n = ids.shape[1]
        return ops.assemble_sequence([self.text_emb.weight, self.visual_emb.weight, self.image_emb.weight], ids.contiguous(),
                                     self._seg[first_pos:first_pos + n].contiguous(),
                                     self._pos_rows()[first_pos:first_pos + n].contiguous())

    def _draw(self, block_logits, filter_thres, temperature, race, name):
        """One token per row from the logits of the position's class block (dalle_artv.py:274-276): top_k over ALL
        total_tokens classes keeps k = int((1 - thres) * total_tokens) of them; the classes outside the block sit at -max,
        so the filter only ever removes block classes when k is smaller than the block."""
        B, n = block_logits.shape
        k_keep = max(int((1 - filter_thres) * self.total_tokens), 1)
        if k_keep < n:
            kept = torch.full_like(block_logits, float('-inf'))
            val, ind = torch.topk(block_logits, k_keep)
            block_logits = kept.scatter_(1, ind, val)
        E = race(name, (B, n)) if race is not None else ops.exponential_like((B, n), block_logits.device)
        tok, _ = ops.sample_race(block_logits.contiguous(), E, None, 0.0, logit_div=temperature, want_y=False)
        return tok.view(B, 1)

    def _sample_cached(self, h, cache, first_pos, filter_thres, temperature, race):
        """The sampling loop over the key/value cache.  One token = [head logits of the image block -> draw -> embedding
        row of the drawn token -> one decode step through the tower]; that chain is captured once (hipGraph) and replayed
        per token: the position lives in a device scalar the step advances, the race variates come from torch's
        graph-safe device generator.  Injected variates (`race`, tests) or a top-k that actually filters run the same
        kernels eagerly."""
        B, dev = h.shape[0], h.device
        c0, c1 = self._allowed_range(self.control_seq_len)
        V = c1 - c0
        lin, ln = self.to_logits[1], self.to_logits[0]
        w_blk, b_blk = self._w16()[c0:c1], lin.bias.detach()[c0:c1].contiguous()
        pos_rows = self._pos_rows().detach().contiguous()
        iemb = self.image_emb.weight.detach()
        sess = self.transformer.decode_session(cache, first_pos, graph=False)
        k_keep = max(int((1 - filter_thres) * self.total_tokens), 1)
        steps = self.target_seq_len
        out = torch.empty(B, steps, dtype=torch.long, device=dev)
        hbuf, logits = h.clone(), torch.empty(B, V, device=dev)
        tok, E = torch.empty(B, dtype=torch.long, device=dev), torch.empty(B, V, device=dev)
        # production: the race variates of the whole loop in one draw (an exponential_ inside the captured step costs its launch and two
        # generator-state fills per replay: 12 us per token); the draw of token n reads block n = position - first_pos.  Capped at
        # 256 MB on top of the KV cache (batch 64 at 1,024 steps x 1,024 codes) and at the 1,024 rows the indexed draw kernel takes: larger
        # calls draw per step (the `E.exponential_()` path below)
        E_all = torch.empty(steps, B, V, device=dev).exponential_() if (race is None and steps * B * V <= (1 << 26) and B <= 1024) else None

        hid = [hbuf]  # the hidden state the next draw reads: the prompt's last position, then the session's output buffer

        def draw(step):
            src = hid[0]
            if self.stable:
                hbuf.copy_(self.norm_by_max(src))
                src = hbuf
            ops.gemv_rows(src, w_blk, b_blk, ln=(ln.weight, ln.bias, ln.eps), round_in=True, out=logits)  # LN + head block
            lg = logits
            if k_keep < V:
                val, ind = torch.topk(lg, k_keep)
                lg = torch.full_like(lg, float('-inf')).scatter_(1, ind, val)
            if E_all is not None:
                ops.sample_race(lg, E_all, None, 0.0, logit_div=temperature, want_y=False, tok_out=tok, step_dev=sess.pos, step0=first_pos)
                return
            if race is not None:
                E.copy_(race(f'tok{step}', (B, V)))
            else:
                E.exponential_()
            ops.sample_race(lg, E, None, 0.0, logit_div=temperature, want_y=False, tok_out=tok)

        def advance():
            # the embedding row of the drawn token (which the same launch files in `out` at column pos - first_pos), then one position
            # through the tower; advances sess.pos
            ops.decode_embed(tok, iemb, pos_rows, sess.pos, sess.x, record=out, record_pos0=first_pos)
            sess._enqueue()
            sess.host_pos += 1  # (the host mirror of the device position: DecodeSession keeps it for step() / token_step() itself)
            hid[0] = sess.y

        graph = None
        use_graph = race is None and k_keep >= V and not self.stable and steps > 4
        # batch 1-2 in production: the whole token (embedding -> tower -> head -> draw) is ONE persistent launch (MMVID_DECODE_TOKEN=0: the
        # launches below).  The first token is drawn from the prompt's hidden state the usual way; every launch then embeds the token drawn
        # last, files it in `out`, and draws the next one.
        if (use_graph and sess.persistent and E_all is not None and V <= 2048 and os.environ.get('MMVID_DECODE_TOKEN', '1') != '0'):
            tk = _lib.DecodeToken()
            tk.tok, tk.table, tk.table_rows, tk.pos_rows, tk.pos_off = tok.data_ptr(), iemb.data_ptr(), iemb.shape[0], pos_rows.data_ptr(), 0
            tk.record, tk.record_ld, tk.record_pos0 = out.data_ptr(), out.stride(0), first_pos
            lnw, lnb = ln.weight.detach(), ln.bias.detach()
            tk.lnf_w, tk.lnf_b, tk.lnf_eps, tk.head_w, tk.head_b, tk.V = lnw.data_ptr(), lnb.data_ptr(), ln.eps, w_blk.data_ptr(), b_blk.data_ptr(), V
            tk.E, tk.e_step_stride, tk.e_pos0, tk.temperature, tk.tok_offset, tk.logits_out = E_all.data_ptr(), B * V, first_pos, temperature, 0, None
            draw(0)
            direct = os.environ.get('MMVID_DECODE_TOKEN_GRAPH', '0') == '0'  # one kernel per token: launched directly (a one-node graph replay costs more)
            # Restart point.  The launch needs its 256 blocks resident together; if the device is shared while it runs, a poll times
            # out, the launch and all later ones on the session's workspace are void, and the failure flag says so.  Every CHECK tokens
            # the flag is read (one sync per ~15 ms of work) and the token to embed next is kept; after a failure the loop goes back to
            # the last verified token and finishes with the launches below (five per layer), which need no co-residency.
            CHECK = getattr(self, '_decode_check_every', 64)
            hook = getattr(self, '_token_step_hook', None)  # tests: called as hook(tokens launched so far, session) after every launch
            good_step, good_tok = 0, tok.clone()
            step = 0
            while step < steps - 1:
                if graph is not None:
                    graph.replay()
                    sess.host_pos += 1
                else:
                    sess.token_step(tk)
                    if step == 1 and not direct:
                        graph = torch.cuda.CUDAGraph()
                        side = torch.cuda.Stream()
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            with torch.cuda.graph(graph, stream=side):
                                sess.token_step(tk)
                        sess.host_pos -= 1  # (the capture enqueued nothing)
                        torch.cuda.current_stream().wait_stream(side)
                step += 1
                if hook is not None:
                    hook(step, sess)
                if step % CHECK == 0 or step == steps - 1:
                    if sess.failed():
                        break
                    good_step = step
                    good_tok.copy_(tok)
            else:
                out[:, steps - 1].copy_(tok)
                return [out[:, i:i + 1] for i in range(steps)]
            # a persistent launch failed somewhere after token `good_step`: go on from there with the separate launches
            tok.copy_(good_tok)
            sess.fall_back(first_pos + good_step)
            graph = None
            for step in range(good_step, steps - 1):
                if step > good_step:
                    draw(step)
                advance()
            draw(steps - 1)
            out[:, steps - 1].copy_(tok)
            return [out[:, i:i + 1] for i in range(steps)]
        for step in range(steps - 1):  # every token but the last: draw it, then run it through the tower
            if graph is not None:
                graph.replay()
                sess.host_pos += 1
                continue
            draw(step)
            advance()
            if use_graph and step == 1:
                # two eager steps have warmed every kernel; capture [draw -> advance] once and replay it
                graph = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    with torch.cuda.graph(graph, stream=side):
                        draw(-1)
                        advance()
                sess.host_pos -= 1  # (the capture enqueued nothing)
                torch.cuda.current_stream().wait_stream(side)
        draw(steps - 1)
        out[:, steps - 1].copy_(tok)
        sess.check()
        return [out[:, i:i + 1] for i in range(steps)]

    def sampling_probs(self, block_logits, filter_thres=0.5, temperature=1.0):
        """The probability vector `_draw` samples from (tests compare it with the reference's full-width expression)."""
        k_keep = max(int((1 - filter_thres) * self.total_tokens), 1)
        if k_keep < block_logits.shape[1]:
            val, ind = torch.topk(block_logits, k_keep)
            block_logits = torch.full_like(block_logits, float('-inf')).scatter_(1, ind, val)
        return F.softmax(block_logits / temperature, dim=-1)

    @torch.no_grad()
    @eval_decorator
    def generate_images(self, text, *, clip=None, visual=None, mask=None, filter_thres=0.5, temperature=1.,
                        erase_visual=False, vc_mode=None, face_mode=None, use_cache=True, _race=None, **kwargs):
        """dalle_artv.py:236-304.  use_cache=True (default): the prompt runs once and every sampled token costs one
        incremental step over the per-layer key/value cache; use_cache=False: the reference's algorithm (the whole
        transformer over the growing prefix per token).  Both draw from the same distribution: softmax over the image
        block of the last position's logits (tests/test_parity_gpu.py compares it with the reference's expression)."""
        tsl = self.text_seq_len
        text = text[:, :tsl]
        B = text.shape[0]
        # visual control tokens (with the erasing the reference applies on every step, dalle_artv.py:464-472) are fixed
        # during sampling: tokenise once
        vis_tok = self._visual_tokens(visual, erase_visual, True, vc_mode, face_mode, None)
        cl = self.control_seq_len
        c0, c1 = self._allowed_range(cl)
        toks = []
        if use_cache:
            cache = self.transformer.new_kv_cache(B, self.total_seq_len, text.device)
            prompt = torch.cat(self._prompt_ids(text, vis_tok), 1)  # <bos> text visual: positions 0 .. cl
            h = self.transformer.prefill(self._embed_rows(prompt, 0), cache)[:, -1, :].contiguous()
            toks = self._sample_cached(h, cache, prompt.shape[1], filter_thres, temperature, _race)
        else:
            image = torch.empty(B, 0, dtype=torch.long, device=text.device)
            for step in range(self.target_seq_len):
                hidden = self._hidden(text, vis_tok, image)[0]
                sample = self._draw(self._logits_rows(hidden[:, -1, :].contiguous(), (c0, c1)), filter_thres, temperature,
                                    _race, f'tok{step}')
                toks.append(sample)
                image = torch.cat((image, sample), dim=-1)
        img_seq = torch.cat(toks, dim=-1).reshape(-1, self.image_seq_len)
        images = self.vae.decode(img_seq)
        if self.num_targets > 1:
            images = images.view(-1, self.num_targets, *images.shape[1:])
        if exists(clip):
            out = torch.cat((text, torch.cat(toks, dim=-1) + self.num_control_tokens), dim=-1)
            return images, clip(out[:, :tsl], images, return_loss=False)
        return images, [], None
