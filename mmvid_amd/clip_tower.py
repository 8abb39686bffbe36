"""Host-side mirror of mmvid_pytorch/transformers/clip_model.py::OpenAICLIPTransformer (520-584) over the
native tower of csrc/tower.hip.  Same constructor meaning, same state_dict keys
(`transformer.resblocks.{i}.{attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.*, ln_1.*, mlp.c_fc.*,
mlp.c_proj.*, ln_2.*}`), same forward contract ([B, L, E] fp32 in/out).  The attention mask is kept as the
predicate the reference's build_attention_mask (561-578) encodes, not as an L x L tensor."""
import ctypes
import os
import math

import torch
from torch import nn

from . import _lib, ops
from .functional import _note_params, _with_param_zeros

TOWER_SHAPES = {  # which_model -> (width, layers, heads), clip_model.py:538-541 with ViT-B/32
    'openai_clip_visual': (768, 12, 12),
    'openai_clip_text': (512, 12, 8),
}
MATRIX_KEYS = ('attn.in_proj_weight', 'attn.out_proj.weight', 'mlp.c_fc.weight', 'mlp.c_proj.weight')


class _Holder(nn.Module):
    """Parameter container (no forward): gives parameters their reference names."""


def _linear_params(out_f, in_f, std):
    h = _Holder()
    h.weight = nn.Parameter(torch.randn(out_f, in_f) * std)
    h.bias = nn.Parameter(torch.zeros(out_f))
    return h


def _ln_params(e):
    h = _Holder()
    h.weight = nn.Parameter(torch.ones(e))
    h.bias = nn.Parameter(torch.zeros(e))
    return h


class ResidualAttentionBlockParams(_Holder):
    def __init__(self, width, layers):
        super().__init__()
        # CLIP.initialize_parameters (clip_model.py:348-378) scales
        proj_std = (width**-0.5) * ((2 * layers)**-0.5)
        attn_std = width**-0.5
        fc_std = (2 * width)**-0.5
        self.attn = _Holder()
        self.attn.in_proj_weight = nn.Parameter(torch.randn(3 * width, width) * attn_std)
        self.attn.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.attn.out_proj = _linear_params(width, width, proj_std)
        self.ln_1 = _ln_params(width)
        self.mlp = _Holder()
        self.mlp.c_fc = _linear_params(4 * width, width, fc_std)
        self.mlp.c_proj = _linear_params(width, 4 * width, proj_std)
        self.ln_2 = _ln_params(width)


class _TowerParams(_Holder):
    def __init__(self, width, layers):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlockParams(width, layers) for _ in range(layers)])


class _TowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tower, keep, *params):
        _note_params(ctx, (x, tower, keep) + params)  # under DistributedDataParallel: see functional._with_param_zeros
        ctx.tower = tower
        y, saved = tower._run_forward(x, keep=keep)  # NB: grad mode is always off inside Function.forward
        ctx.saved_arena = saved
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        g = gy.contiguous().clone()  # updated in place into dL/dx
        ctx.tower._run_backward(g, ctx.saved_arena, ctx.shape)
        ctx.saved_arena = None
        return _with_param_zeros(ctx, (g.view(ctx.shape), None, None) + (None, ) * (len(ctx.needs_input_grad) - 3))


class DecodeSession:
    """See OpenAICLIPTransformer.decode_session.  step(x_new [B, width]) -> hidden [B, width] (a static buffer that
    the next step overwrites)."""

    def __init__(self, tower, kv_cache, first_pos, graph, fused=True):
        self.tower, self.cache, self.fused = tower, kv_cache, fused
        B, dev = kv_cache.shape[1], kv_cache.device
        self.cfg = tower._cfg(B, kv_cache.shape[2])
        self.layers, self._keep = tower._layer_structs(False)
        _, self.scratch = tower._workspace(self.cfg, dev, False)
        self.x = torch.zeros(B, tower.width, device=dev, dtype=torch.float32)
        self.y = torch.zeros_like(self.x)
        self.pos = torch.tensor([first_pos], dtype=torch.int32, device=dev)
        self.host_pos = int(first_pos)  # host mirror of the device position (explicit: not derived from the number of step() calls)
        self.graph, self.want_graph, self.calls = None, graph, 0
        self._slice_cfgs = {}
        # the whole step as one launch of 256 co-resident blocks (csrc/decode_persistent.hip) where the tower / batch / device allow it;
        # MMVID_DECODE_PERSISTENT=0 (or fused='launches') keeps the five-launches-per-layer form
        lib = _lib.load()
        self.persistent = bool(fused is True and os.environ.get('MMVID_DECODE_PERSISTENT', '1') != '0'
                               and lib.mmvid_tower_decode_persistent_supported(ctypes.byref(self.cfg), kv_cache.shape[2]))
        self.ws = torch.zeros(lib.mmvid_tower_decode_persistent_workspace_bytes(B) // 8, dtype=torch.int64, device=dev) if self.persistent else None

    def _enqueue(self):
        B, E = self.x.shape[0], self.x.shape[-1]
        if self.persistent:
            _lib.call('mmvid_tower_decode_persistent', ctypes.byref(self.cfg), self.layers, ops._p(self.x), ops._p(self.y),
                      ops._p(self.cache), self.cache.shape[2], ops._p(self.pos), 0, 1, ops._p(self.ws), ops._stream())  # (advances pos itself)
            return
        Lmax = self.cache.shape[2]
        if self.fused and E <= 768:
            # slices of up to 64 sequences through the decode kernels of csrc/decode.hip (3..64 rows: the linear layers on the matrix pipe,
            # the weights streamed once per slice; the M = B corner of the training GEMM took 2.2 ms per token at batch 16); the last
            # slice's last launch advances the position
            per = 64 if E in (512, 768) else 16  # (the matrix-pipe linear layers are instantiated for the CLIP widths; others: 16 rows)
            for b0 in range(0, B, per):
                nb = min(per, B - b0)
                cfg = self.cfg if nb == B else self._slice_cfgs.setdefault(nb, self.tower._cfg(nb, Lmax))
                _lib.call('mmvid_tower_decode_fused_slice', ctypes.byref(cfg), self.layers, self.x[b0:].data_ptr(), self.y[b0:].data_ptr(),
                          self.cache[0, b0].data_ptr(), Lmax, B, ops._p(self.pos), 0, int(b0 + per >= B), ops._p(self.scratch), ops._stream())
            return
        _lib.call('mmvid_tower_decode', ctypes.byref(self.cfg), self.layers, ops._p(self.x), ops._p(self.y),
                  ops._p(self.cache), Lmax, ops._p(self.pos), 0, ops._p(self.scratch), ops._stream())
        self.pos.add_(1)

    def token_step(self, tk):
        """The sampler's whole token in one launch (csrc/decode_persistent.hip: embedding of the token drawn last -> tower step -> head ->
        draw of the next token -> position + 1).  tk: a filled _lib.DecodeToken (the caller keeps its tensors alive).  Persistent sessions only."""
        assert self.persistent
        _lib.call('mmvid_artv_token_step_persistent', ctypes.byref(self.cfg), self.layers, ctypes.byref(tk), ops._p(self.y), ops._p(self.cache),
                  self.cache.shape[2], ops._p(self.pos), ops._p(self.ws), ops._stream())
        self.host_pos += 1

    def failed(self):
        """True when a poll of the persistent step has timed out on this session's workspace (its 256 blocks were not resident together:
        something else held part of the device); every later launch on the workspace returns at once.  One device read (a sync)."""
        return bool(self.persistent and int(self.ws[1]) != 0)

    def fall_back(self, pos):
        """Leave the persistent form for good: this session continues from position `pos` with the five-launch step (the key/value
        entries at and beyond `pos` are rewritten by the steps that follow)."""
        self.persistent, self.graph = False, None
        self.fell_back = getattr(self, 'fell_back', 0) + 1
        self.pos.fill_(int(pos))
        self.host_pos = int(pos)

    def check(self):
        """Raise if the persistent step failed and nobody recovered from it (`step(verify=True)` and the ART-V sampling loop do, by
        repeating the affected positions with the launch-per-layer step)."""
        if self.failed():
            raise _lib.MMVIDError('persistent decode step: a poll timed out (the device was shared while it ran); results are invalid. '
                                  'Set MMVID_DECODE_PERSISTENT=0 to use the five-launch step.')

    @torch.no_grad()
    def step(self, x_new, verify=False):
        """One position.  Asynchronous by default: steps can be pipelined / replayed without a host read; a persistent session's caller
        calls failed() / check() at its own restart points (the ART-V sampler: every 64 tokens).  verify=True (persistent sessions only)
        reads the failure flag after the step -- one device sync per token -- and repeats a step whose polls timed out with the
        launch-per-layer form, staying there: the returned hidden state is then always valid."""
        self.x.copy_(x_new)
        self.calls += 1
        if self.want_graph and self.graph is None and self.calls == 2:
            # second step: capture (the first one ran eagerly and warmed every kernel variant up)
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    self._enqueue()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = g
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
        if verify and self.persistent and self.failed():
            self.fall_back(self.host_pos)
            self._enqueue()
        self.host_pos += 1
        return self.y


class OpenAICLIPTransformer(nn.Module):
    def __init__(self, seq_len=0, which_model='openai_clip_text', model_path=None, causal=True, mask_type='causal',
                 mask_kwargs=None, layers=None, width=None, heads=None):
        super().__init__()
        if not which_model.startswith('openai_clip'):
            raise NotImplementedError(which_model)  # as dalle_bert.py:406-407
        w, l, h = TOWER_SHAPES[which_model]
        self.width, self.layers, self.heads = width or w, layers or l, heads or h
        self.debug_keep_saved = False
        assert self.width == 64 * self.heads, 'head_dim must be 64 (CLIP towers)'
        self.context_length = seq_len
        self.causal = causal
        self.transformer = _TowerParams(self.width, self.layers)
        self.mask_spec = self.build_attention_mask(seq_len, mask_type, **(mask_kwargs or {})) if causal else None
        if model_path is not None:
            self.load_clip_checkpoint(model_path, which_model)
        self._shadow = None  # flat bf16 copy of the matrix weights
        self._shadow_key = None
        self._scratch = None
        self._scratch_retired = []
        self.backward_chunk_layers = int(os.environ.get('MMVID_BWD_CHUNK', 0))  # the backward returns to the host every N layers (0: planned, see backward_chunks) ...
        self.on_layers_done = None      # ... and calls this (first_layer) so gradient exchange can overlap

    # ---- reference API -----------------------------------------------------------------------------
    @staticmethod
    def build_attention_mask(context_length, mask_type='causal', **kwargs):
        """clip_model.py:561-578 as a predicate: 'causal' or ('rows', [(i, i) for i in index])."""
        if mask_type == 'causal':
            return 'causal'
        if mask_type == 'mask_prev':
            idx = list(kwargs['index'])
            assert len(idx) <= 2, 'the kernels carry at most two restricted rows (BERT uses [ST1],[VID])'
            return ('rows', [(int(i), int(i)) for i in idx])
        raise NotImplementedError(mask_type)

    def dense_attention_mask(self, L=None):
        """The additive [L, L] float mask the reference would have built (for tests / inspection)."""
        L = L or self.context_length
        if self.mask_spec is None:
            return None
        if self.mask_spec == 'causal':
            return torch.full((L, L), float('-inf')).triu_(1)
        m = torch.zeros(L, L)
        for r, c in self.mask_spec[1]:
            m[r, :c] = float('-inf')
        return m

    def load_clip_checkpoint(self, path, which_model):
        """clip_model.py:535-559: pull one tower out of OpenAI's TorchScript archive (fp16 weights -> fp32)."""
        sd = torch.jit.load(path, map_location='cpu').state_dict()
        prefix = 'transformer.' if which_model == 'openai_clip_text' else 'visual.transformer.'
        own = {k[len(prefix):]: v.float() for k, v in sd.items() if k.startswith(prefix + 'resblocks.')}
        self.transformer.load_state_dict(own)

    def forward(self, x, **kwargs):
        assert x.dim() == 3 and x.shape[-1] == self.width
        keep = torch.is_grad_enabled() and (x.requires_grad or self._any_trainable())
        return _TowerFn.apply(x, self, keep, *self.parameters())

    # ---- incremental decoding (causal towers; SURVEY next-row N1) -----------------------------------------
    def new_kv_cache(self, B, max_len, device):
        """[layers, B, max_len, 2*width] bf16: per layer and position the key row followed by the value row."""
        assert self.mask_spec == 'causal', 'incremental decoding needs the causal mask'
        return torch.empty(self.layers, B, max_len, 2 * self.width, device=device, dtype=torch.bfloat16)

    @torch.no_grad()
    def prefill(self, x, kv_cache):
        """Causal forward over the prompt x [B, L, width] that also stores every layer's keys / values."""
        x = ops._chk(x.contiguous(), torch.float32, 'tower input')
        B, L, _ = x.shape
        cfg = self._cfg(B, L)
        layers, _keep = self._layer_structs(False)
        _, scratch = self._workspace(cfg, x.device, False)
        y = torch.empty_like(x)
        _lib.call('mmvid_tower_prefill', ctypes.byref(cfg), layers, ops._p(x), ops._p(y), ops._p(kv_cache),
                  kv_cache.shape[2], ops._p(scratch), ops._stream())
        return y

    @torch.no_grad()
    def decode_step(self, x_new, kv_cache, pos, pos_dev=None):
        """One new position: x_new [B, width] at index `pos` (or the int32 device scalar pos_dev) -> [B, width]."""
        x_new = ops._chk(x_new.contiguous(), torch.float32, 'tower input')
        B = x_new.shape[0]
        cfg = self._cfg(B, kv_cache.shape[2])  # scratch sized like a forward over the whole cache length
        layers, _keep = self._layer_structs(False)
        _, scratch = self._workspace(cfg, x_new.device, False)
        y = torch.empty_like(x_new)
        _lib.call('mmvid_tower_decode', ctypes.byref(cfg), layers, ops._p(x_new), ops._p(y), ops._p(kv_cache),
                  kv_cache.shape[2], ops._p(pos_dev), int(pos), ops._p(scratch), ops._stream())
        return y

    def decode_session(self, kv_cache, first_pos, graph=True, fused=True):
        """A sampling loop's view of decode_step: argument structs and buffers are built once, the position lives in a
        device scalar that the step itself advances, and (graph=True) the step is captured once and replayed."""
        return DecodeSession(self, kv_cache, first_pos, graph, fused)

    # ---- native plumbing ---------------------------------------------------------------------------
    def _any_trainable(self):
        return any(p.requires_grad for p in self.parameters())

    def _matrix_params(self):
        for blk in self.transformer.resblocks:
            yield blk.attn.in_proj_weight
            yield blk.attn.out_proj.weight
            yield blk.mlp.c_fc.weight
            yield blk.mlp.c_proj.weight

    def attach_shadow(self, views):
        """Engine hook: `views` = list of bf16 tensors (one per matrix param, in _matrix_params order) that an
        external fused optimiser keeps equal to bf16(param)."""
        self._shadow = list(views)
        self.mark_shadow_fresh()

    def mark_shadow_fresh(self):
        self._shadow_key = tuple((p._version, p.data_ptr()) for p in self._matrix_params())

    def _sync_shadow(self):
        ps = list(self._matrix_params())
        key = tuple((p._version, p.data_ptr()) for p in ps)
        if self._shadow is None or self._shadow[0].device != ps[0].device:
            self._shadow = [torch.empty(p.shape, device=p.device, dtype=torch.bfloat16) for p in ps]
            self._shadow_key = None
        if key != self._shadow_key:
            for p, s in zip(ps, self._shadow):
                ops.cast_bf16(p.detach().contiguous(), s)
            self._shadow_key = key
        return self._shadow

    def _cfg(self, B, L):
        c = _lib.TowerCfg()
        c.B, c.L, c.E, c.H, c.F, c.layers = B, L, self.width, self.heads, 4 * self.width, self.layers
        c.mask_mode, c.r0, c.c0, c.r1, c.c1 = ops._mask_args(self.mask_spec)
        c.ln_eps = 1e-5
        return c

    def _layer_structs(self, with_grads):
        sh = self._sync_shadow()
        arr = (_lib.TowerLayer * self.layers)()
        keep = []

        def ptr(t):
            keep.append(t)
            return t.data_ptr()

        def gptr(p):
            if not with_grads or not p.requires_grad:
                return None
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            return p.grad.data_ptr()

        for i, blk in enumerate(self.transformer.resblocks):
            a = arr[i]
            a.ln1_w, a.ln1_b = ptr(blk.ln_1.weight), ptr(blk.ln_1.bias)
            a.ln2_w, a.ln2_b = ptr(blk.ln_2.weight), ptr(blk.ln_2.bias)
            a.in_w, a.in_b = ptr(sh[4 * i]), ptr(blk.attn.in_proj_bias)
            a.out_w, a.out_b = ptr(sh[4 * i + 1]), ptr(blk.attn.out_proj.bias)
            a.fc_w, a.fc_b = ptr(sh[4 * i + 2]), ptr(blk.mlp.c_fc.bias)
            a.pj_w, a.pj_b = ptr(sh[4 * i + 3]), ptr(blk.mlp.c_proj.bias)
            a.g_ln1_w, a.g_ln1_b = gptr(blk.ln_1.weight), gptr(blk.ln_1.bias)
            a.g_ln2_w, a.g_ln2_b = gptr(blk.ln_2.weight), gptr(blk.ln_2.bias)
            a.g_in_w, a.g_in_b = gptr(blk.attn.in_proj_weight), gptr(blk.attn.in_proj_bias)
            a.g_out_w, a.g_out_b = gptr(blk.attn.out_proj.weight), gptr(blk.attn.out_proj.bias)
            a.g_fc_w, a.g_fc_b = gptr(blk.mlp.c_fc.weight), gptr(blk.mlp.c_fc.bias)
            a.g_pj_w, a.g_pj_b = gptr(blk.mlp.c_proj.weight), gptr(blk.mlp.c_proj.bias)
        return arr, keep

    def _workspace(self, cfg, device, keep):
        sb, cb = ctypes.c_int64(), ctypes.c_int64()
        _lib.call('mmvid_tower_workspace', ctypes.byref(cfg), ctypes.byref(sb), ctypes.byref(cb))
        # grow-only scratch: a sampler that calls with a longer sequence every step must not reallocate ~1 GB each time
        if self._scratch is None or self._scratch.device != torch.device(device) or self._scratch.numel() < cb.value:
            if self._scratch is not None:
                self._scratch_retired.append(self._scratch)  # captured graphs / decode sessions may still point into it
            self._scratch = torch.empty(cb.value, device=device, dtype=torch.uint8)
        saved = torch.empty(sb.value, device=device, dtype=torch.uint8) if keep else None
        return saved, self._scratch

    def _run_forward(self, x, keep):
        x = ops._chk(x.contiguous(), torch.float32, 'tower input')
        B, L, _ = x.shape
        cfg = self._cfg(B, L)
        layers, _keep = self._layer_structs(False)
        saved, scratch = self._workspace(cfg, x.device, keep)
        y = torch.empty_like(x)
        _lib.call('mmvid_tower_forward', ctypes.byref(cfg), layers, ops._p(x), ops._p(y), ops._p(saved), ops._p(scratch),
                  ops._stream())
        if self.debug_keep_saved:  # debugging aid (tools/stress_nan2.py): the activation arena of the latest forward
            self._last_saved = saved  # (layout: csrc/tower.hip saved_layout); holding it changes the allocation pattern
        return y, saved

    def backward_chunks(self):
        """[(first_layer, end_layer)] in the order the backward walks them: the native layer loop returns to the host after
        each chunk so that the finished layers' gradients can go on the wire (engine.FlatTrainer.layers_done)."""
        if self.on_layers_done is None:
            return [(0, self.layers)]
        if self.backward_chunk_layers <= 0:
            return self._planned_chunks()
        step = self.backward_chunk_layers
        out, hi = [], self.layers
        while hi > 0:
            lo = max(0, hi - step)
            out.append((lo, hi))
            hi = lo
        return out

    def _planned_chunks(self):
        """Two backward calls whose grouped weight-gradient launches waste the fewest CU rounds.  A call over n layers launches
        n * t output tiles of 256 x 128 (t = 216 for width 768 / 3072) on 256 CUs; what is lost is the unfilled part of the last round.
        Three-layer calls (rounds 1-3) lost 1.9 rounds per backward (0.34 ms of the +0.42 ms a forced exchange cost); 7 + 5 layers lose
        0.87 -- what the single launch of the exchange-free step loses (0.875) -- and the first call's 58 % of the gradients go on the
        wire while five layers are still being differentiated."""
        if getattr(self, '_chunk_plan', None) is None or self._chunk_plan[0] != self.layers:
            E = self.width
            F = 4 * E
            t = sum(-(-n // 256) * -(-k // 128) for n, k in ((3 * E, E), (E, E), (F, E), (E, F)))
            waste = lambda n: (-(-n * t // 256)) - n * t / 256.0
            best, plan = None, [(0, self.layers)]
            for a in range(3, self.layers - 2):  # a = layers of the FIRST call (the top of the tower); both calls >= 3 layers
                w = waste(a) + waste(self.layers - a) + 0.02 * abs(a - 0.6 * self.layers)  # (ties: the first call a little larger)
                if best is None or w < best:
                    best, plan = w, [(self.layers - a, self.layers), (0, self.layers - a)]
            self._chunk_plan = (self.layers, plan)
        return self._chunk_plan[1]

    def _run_backward(self, g, saved, shape):
        if saved is None:
            raise _lib.MMVIDError('tower backward without saved activations (forward ran under no_grad)')
        B, L, _ = shape
        cfg = self._cfg(B, L)
        layers, _keep = self._layer_structs(True)
        _, scratch = self._workspace(cfg, g.device, False)
        sb = ctypes.c_int64()
        _lib.call('mmvid_tower_workspace', ctypes.byref(cfg), ctypes.byref(sb), None)
        if sb.value != saved.numel():  # the arena's layout (kept dY tensors) must not change in between
            raise _lib.MMVIDError(f'tower backward: the saved arena holds {saved.numel()} bytes, the current layout needs {sb.value} '
                                  '(the library changed between forward and backward)')
        per_layer = saved.numel() // self.layers
        for lo, hi in self.backward_chunks():
            sub = self._cfg(B, L)
            sub.layers = hi - lo
            lp = ctypes.cast(ctypes.byref(layers, lo * ctypes.sizeof(_lib.TowerLayer)), ctypes.POINTER(_lib.TowerLayer))
            _lib.call('mmvid_tower_backward', ctypes.byref(sub), lp, ops._p(g),
                      ctypes.c_void_p(saved.data_ptr() + lo * per_layer), ops._p(scratch), ops._stream())
            if self.on_layers_done is not None:
                self.on_layers_done(lo)
