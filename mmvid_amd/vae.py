"""Host-side mirror of mmvid_pytorch/vae.py::VQGanVAE1024 (15-71) -> taming VQModel (vqgan.py:16-75) over the HIP
kernels: same constructor, attributes (`image_size` is mutated by the driver, train.py:183; `num_layers`,
`num_tokens`), methods (`get_codebook_indices`, `decode`, `decode_train`) and the same state_dict keys as
`model.encoder.* / model.decoder.* / model.quantize.embedding.weight / model.quant_conv.* / model.post_quant_conv.*`
(taming/modules/diffusionmodules/model.py:363-582), so published `vqgan.1024.model.ckpt` files load.

Execution (csrc/conv.hip, norm.hip, vq.hip): activations are NHWC; the residual stream is fp32, every conv
input is bf16 (MFMA, fp32 accumulate); GroupNorm statistics are fp32.  The frozen weights are re-laid-out once
into [Cout][ky][kx][Cin] bf16 (cached, refreshed when a parameter changes)."""
from math import sqrt

import os

import torch
from torch import nn

from . import _lib, ops

bf16, f32 = torch.bfloat16, torch.float32

# mmvid_pytorch/data/vqgan.1024.config.yml
DEFAULT_DDCONFIG = dict(double_z=False, z_channels=256, resolution=256, in_channels=3, out_ch=3, ch=128,
                        ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16, ), dropout=0.0)


class _H(nn.Module):
    """parameter holder"""


def _conv(cin, cout, k):
    h = _H()
    h.weight = nn.Parameter(torch.empty(cout, cin, k, k))
    h.bias = nn.Parameter(torch.empty(cout))
    nn.init.kaiming_uniform_(h.weight, a=sqrt(5))
    bound = 1 / sqrt(cin * k * k)
    nn.init.uniform_(h.bias, -bound, bound)
    return h


def _norm(c):
    h = _H()
    h.weight = nn.Parameter(torch.ones(c))
    h.bias = nn.Parameter(torch.zeros(c))
    return h


def _resblock(cin, cout):
    h = _H()
    h.norm1, h.conv1, h.norm2, h.conv2 = _norm(cin), _conv(cin, cout, 3), _norm(cout), _conv(cout, cout, 3)
    if cin != cout:
        h.nin_shortcut = _conv(cin, cout, 1)
    return h


def _attnblock(c):
    h = _H()
    h.norm, h.q, h.k, h.v, h.proj_out = _norm(c), _conv(c, c, 1), _conv(c, c, 1), _conv(c, c, 1), _conv(c, c, 1)
    return h


def _mid(c):
    h = _H()
    h.block_1, h.attn_1, h.block_2 = _resblock(c, c), _attnblock(c), _resblock(c, c)
    return h


class _Encoder(_H):  # model.py:363-437
    def __init__(self, *, ch, ch_mult, num_res_blocks, attn_resolutions, in_channels, resolution, z_channels,
                 double_z=False, **_):
        super().__init__()
        self.conv_in = _conv(in_channels, ch, 3)
        cur = resolution
        in_mult = (1, ) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for lvl in range(len(ch_mult)):
            d = _H()
            d.block, d.attn = nn.ModuleList(), nn.ModuleList()
            bin_, bout = ch * in_mult[lvl], ch * ch_mult[lvl]
            for _b in range(num_res_blocks):
                d.block.append(_resblock(bin_, bout))
                bin_ = bout
                if cur in attn_resolutions:
                    d.attn.append(_attnblock(bin_))
            if lvl != len(ch_mult) - 1:
                d.downsample = _H()
                d.downsample.conv = _conv(bin_, bin_, 3)
                cur //= 2
            self.down.append(d)
        self.mid = _mid(bin_)
        self.norm_out = _norm(bin_)
        self.conv_out = _conv(bin_, 2 * z_channels if double_z else z_channels, 3)


class _Decoder(_H):  # model.py:469-549
    def __init__(self, *, ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels, **_):
        super().__init__()
        n = len(ch_mult)
        bin_ = ch * ch_mult[n - 1]
        cur = resolution // 2**(n - 1)
        self.conv_in = _conv(z_channels, bin_, 3)
        self.mid = _mid(bin_)
        ups = []
        for lvl in reversed(range(n)):
            u = _H()
            u.block, u.attn = nn.ModuleList(), nn.ModuleList()
            bout = ch * ch_mult[lvl]
            for _b in range(num_res_blocks + 1):
                u.block.append(_resblock(bin_, bout))
                bin_ = bout
                if cur in attn_resolutions:
                    u.attn.append(_attnblock(bin_))
            if lvl != 0:
                u.upsample = _H()
                u.upsample.conv = _conv(bin_, bin_, 3)
                cur *= 2
            ups.insert(0, u)
        self.up = nn.ModuleList(ups)
        self.norm_out = _norm(bin_)
        self.conv_out = _conv(bin_, out_ch, 3)


class VQModel(_H):  # vqgan.py:16-53
    def __init__(self, ddconfig, n_embed, embed_dim, **_):
        super().__init__()
        self.ddconfig = dict(ddconfig)
        self.encoder = _Encoder(**ddconfig)
        self.decoder = _Decoder(**ddconfig)
        self.quantize = _H()
        self.quantize.embedding = nn.Embedding(n_embed, embed_dim)
        self.quantize.embedding.weight.data.uniform_(-1.0 / n_embed, 1.0 / n_embed)  # quantize.py:254
        self.quant_conv = _conv(ddconfig['z_channels'], embed_dim, 1)
        self.post_quant_conv = _conv(embed_dim, ddconfig['z_channels'], 1)


# A/B switches for the planner's fusions (diagnostics; both on by default)
_FUSE_GN = os.environ.get('MMVID_FUSE_GN', '1') != '0'
_DUAL_OUT = os.environ.get('MMVID_DUAL_OUT', '1') != '0'
_STRIP = os.environ.get('MMVID_CONV_STRIP', '1') != '0'
_SPLITK = os.environ.get('MMVID_CONV_SPLITK', '1') != '0'
_FUSE_QKV = os.environ.get('MMVID_FUSE_QKV', '1') != '0'  # AttnBlock q|k|v as one 1x1 conv (bf16 operator only)


def _pow2_at_least8(c):
    p = 8
    while p < c:
        p *= 2
    return p


class _Buf:
    """Planned tensor: a byte range of the arena.  Dropping the last reference returns the range to the planner's
    free list, which is exactly the liveness rule a single in-order stream needs."""

    def __init__(self, pl, off, nbytes, shape, dtype):
        self.pl, self.off, self.nbytes, self.shape, self.dtype = pl, off, nbytes, shape, dtype

    def __del__(self):
        if self.pl is not None and self.pl.recording:
            self.pl.free.append((self.off, self.nbytes))


class _Plan:
    def __init__(self, ops_arr, arena, patches, kept):
        self.ops, self.arena, self.patches, self.kept = ops_arr, arena, patches, kept

    def run(self, ext_in, ext_out):
        for i, field, name in self.patches:
            src = ext_in if field == 'ext_in' else ext_out
            setattr(self.ops[i], field, src[name].data_ptr())
        _lib.call('mmvid_vqgan_run', self.ops, len(self.ops), ops._p(self.arena), ops._stream())


class _Planner:
    OP_IMG, OP_CONV, OP_GN, OP_CAST, OP_ATTN, OP_VQ, OP_GATHER, OP_NCHW, OP_EXT = range(9)
    STRICT = 16  # MMVID_VQFLAG_STRICT: the fp32-accurate operator (csrc/strict.hip); all planned tensors are fp32
    SPLIT = 64   # MMVID_VQFLAG_SPLIT: the bf16-pair operator; planned tensors are fp32 or pair planes [2][n,h,w,c] bf16
    F16 = 128    # MMVID_VQFLAG_F16 (with SPLIT): a GroupNorm that writes one fp16 plane / the strip convolution that reads it

    def __init__(self, vae, strict=False, stream16=False, f16_side=0):
        self.vae, self.ops, self.free, self.top, self.recording = vae, [], [], 0, True
        self.patches, self.kept = [], {}
        self.split = strict == 'split'
        # split operator only: 3x3 stride-1 convolutions on maps of at least f16_side x f16_side pixels that read a GroupNorm output run
        # as ONE product of fp16 operands (vae.strict = 'mixed'); 0 = every convolution is the bf16-pair operator
        self.f16_side = int(f16_side) if self.split else 0
        self.strict = bool(strict) and not self.split
        # bf16 operator only: the residual stream between blocks is stored as bf16 as well (no fp32 activation leaves a conv except
        # the VQ rows and the decoded image); the exact operators keep their fp32 streams
        self.stream16 = bool(stream16) and not self.strict and not self.split

    # ---- split operator (vae.strict = 'split'): fp32 tensors between ops, every conv input a bf16 pair ----------------------
    def _planes(self, x):
        """fp32 tensor -> pair planes (a no-op for a tensor that already is one)."""
        if getattr(x, 'is_planes', False):
            return x
        n, h, wd, c = x.shape
        assert x.dtype == f32 and c % 8 == 0, x.shape
        out = self.alloc((2, n, h, wd, c), bf16)
        out.is_planes, out.shape = True, (n, h, wd, c)
        self._op(op=self.OP_CAST, N=n, H=h, W=wd, C=c, flags=self.SPLIT, in0=x.off, out_bf16=out.off)
        return out

    def _conv_split(self, x, holder, mode, residual, clamp01, feeds_gn=False, planes_only=False):
        x = self._planes(x)
        w3, b, _ = self.vae._cw_split(holder)
        n, h, wd, cin = x.shape
        assert cin == w3.shape[3], (x.shape, w3.shape)
        ho, wo = (h // 2, wd // 2) if mode == 1 else ((2 * h, 2 * wd) if mode == 2 else (h, wd))
        cout = w3.shape[0]
        assert residual is None or residual.dtype == f32
        flags = self.SPLIT | (2 if clamp01 else 0)
        scratch, ws = -1, None
        strip = _STRIP and mode == 0 and not clamp01 and bool(_lib.load().mmvid_conv3x3_strip_supported(h, wd, cin, cout))
        if strip:
            flags |= 8
        # planes_only: the result's only reader is another pair-operator convolution (a level's last tensor in front of its Downsample):
        # the strip kernel's epilogue stores the bf16 pair itself -- no fp32 store, no cast pass (the same planes bit for bit)
        planes_only = planes_only and strip and not feeds_gn
        if planes_only:
            out = self.alloc((2, n, ho, wo, cout), bf16)
            out.is_planes, out.shape = True, (n, ho, wo, cout)
        else:
            out = self.alloc((n, ho, wo, cout), f32)
        gn_op = getattr(x, 'gn_op', None)
        if (strip and gn_op is not None and self.f16_side and min(h, wd) >= self.f16_side and holder.weight.shape[2] == 3
                and getattr(x, 'pair_readers', 0) == 0):
            # the fp16 form: this convolution is the FIRST reader of that GroupNorm's output (a residual block's norm -> conv), so the
            # GroupNorm op already planned is switched to its fp16 output (first plane of the same buffer) and the weights are fp16
            gn_op.flags |= self.F16
            x.f16_plane = True
            flags |= self.F16
            w3 = self.vae._cw_f16(holder)[0]
        else:  # a pair-plane reader: the buffer must still hold (hi, lo) planes, and no later reader may turn it into an fp16 plane
            assert not getattr(x, 'f16_plane', False), 'a GroupNorm output switched to fp16 has a second reader that needs bf16 pairs'
            if gn_op is not None:
                x.pair_readers = getattr(x, 'pair_readers', 0) + 1
        if feeds_gn and _FUSE_GN and (ho * wo) % 128 == 0 and cout % 128 == 0:  # the epilogue emits the GroupNorm partial sums
            out.gn_stats = self._gn_stats(n, ho * wo, cout)
            out.gn_stats.blocks64 = strip
            flags |= 4
            scratch = out.gn_stats.off
        elif not strip and _SPLITK and mode == 0 and ho * wo <= 64 and 9 * cin >= 2304 and cout % 4 == 0:
            ws = self.alloc((4 * n * ho * wo * cout, ), f32)
            flags |= 32
            scratch = ws.off
        self._op(op=self.OP_CONV, mode=mode, N=n, H=h, W=wd, C=cin, Cout=cout, flags=flags, in0=x.off,
                 in1=residual.off if residual is not None else -1, out_f32=-1 if planes_only else out.off,
                 out_bf16=out.off if planes_only else -1, scratch=scratch, w=w3.data_ptr(), b=b.data_ptr())
        del ws
        return out

    # arena allocation: first fit in the free list, else bump
    def alloc(self, shape, dtype):
        n = 1
        for d in shape:
            n *= d
        nbytes = (n * (2 if dtype == bf16 else 4) + 255) // 256 * 256
        for i, (off, sz) in enumerate(self.free):
            if sz >= nbytes:
                if sz > nbytes:
                    self.free[i] = (off + nbytes, sz - nbytes)
                else:
                    self.free.pop(i)
                return _Buf(self, off, nbytes, tuple(shape), dtype)
        off = self.top
        self.top += nbytes
        return _Buf(self, off, nbytes, tuple(shape), dtype)

    def _op(self, **kw):
        o = _lib.VqganOp()
        o.in0 = o.in1 = o.in2 = o.out_bf16 = o.out_f32 = o.scratch = -1
        for k, v in kw.items():
            setattr(o, k, v)
        self.ops.append(o)
        return len(self.ops) - 1

    def image(self, n, s):
        if self.split:
            out = self.alloc((2, n, s, s, 8), bf16)
            out.is_planes, out.shape = True, (n, s, s, 8)
            i = self._op(op=self.OP_IMG, N=n, H=s, W=s, C=3, out_bf16=out.off, flags=self.SPLIT)
            self.patches.append((i, 'ext_in', 'img'))
            return out
        if self.strict:
            out = self.alloc((n, s, s, 4), f32)
            i = self._op(op=self.OP_IMG, N=n, H=s, W=s, C=3, out_f32=out.off, flags=self.STRICT)
            self.patches.append((i, 'ext_in', 'img'))
            return out
        out = self.alloc((n, s, s, 8), bf16)
        i = self._op(op=self.OP_IMG, N=n, H=s, W=s, C=3, out_bf16=out.off)
        self.patches.append((i, 'ext_in', 'img'))
        return out

    def conv(self, x, holder, mode, residual=None, out32=False, clamp01=False, feeds_gn=False, also_bf16=False, keep32=False):
        """feeds_gn: a GroupNorm reads this output next -> the epilogue also emits its partial statistics (when the
        shape allows), into a stats area that lives as long as the output buffer.
        also_bf16 (with out32): the epilogue stores a bf16 copy too (`out.bf16`), instead of a later cast pass.
        keep32: fp32 output even with a bf16 residual stream (the VQ rows, the decoded image)."""
        if self.split:
            return self._conv_split(x, holder, mode, residual, clamp01, feeds_gn, planes_only=not out32 and not keep32)
        if self.stream16 and not keep32:
            out32 = also_bf16 = False
        w, b, _ = self.vae._cw(holder, self.strict)
        n, h, wd, cin = x.shape
        assert x.dtype == (f32 if self.strict else bf16) and cin == w.shape[2], (x.shape, w.shape)
        ho, wo = (h // 2, wd // 2) if mode == 1 else ((2 * h, 2 * wd) if mode == 2 else (h, wd))
        cout = w.shape[0]
        if self.strict:
            out = self.alloc((n, ho, wo, cout), f32)
            self._op(op=self.OP_CONV, mode=mode, N=n, H=h, W=wd, C=cin, Cout=cout,
                     flags=self.STRICT | (2 if clamp01 else 0), in0=x.off,
                     in1=residual.off if residual is not None else -1, out_f32=out.off, w=w.data_ptr(), b=b.data_ptr())
            return out
        out = self.alloc((n, ho, wo, cout), f32 if out32 else bf16)
        flags = (1 if (residual is not None and residual.dtype == f32) else 0) | (2 if clamp01 else 0)
        # 3x3 stride-1 layers at 32x32 and above run in strip form (csrc/conv_strip.hip); the rule is geometry only
        strip = _STRIP and mode == 0 and not clamp01 and bool(_lib.load().mmvid_conv3x3_strip_supported(h, wd, cin, cout))
        if strip:
            flags |= 8
        scratch = -1
        if feeds_gn and _FUSE_GN and (ho * wo) % 128 == 0 and cout % 128 == 0:
            out.gn_stats = self._gn_stats(n, ho * wo, cout)
            out.gn_stats.blocks64 = strip
            flags |= 4
            scratch = out.gn_stats.off
        ws = None
        if _SPLITK and not strip and scratch < 0 and mode == 0 and ho * wo <= 64 and 9 * cin >= 2304 and cout % 4 == 0:
            # deep 3x3 layer on an 8x8 map: its 128x128 output tiles alone cover a fraction of the chip -> split-K by 4 through
            # an fp32 workspace, fixed-order reduce (the rule is geometry only, like every kernel choice of the encoder)
            ws = self.alloc((4 * n * ho * wo * cout, ), f32)
            flags |= 32
            scratch = ws.off
        o16 = out.off if not out32 else -1
        if out32 and also_bf16 and _DUAL_OUT:
            out.bf16 = self.alloc((n, ho, wo, cout), bf16)
            o16 = out.bf16.off
        self._op(op=self.OP_CONV, mode=mode, N=n, H=h, W=wd, C=cin, Cout=cout, flags=flags, in0=x.off,
                 in1=residual.off if residual is not None else -1, out_bf16=o16,
                 out_f32=out.off if out32 else -1, scratch=scratch, w=w.data_ptr(), b=b.data_ptr())
        del ws  # the workspace returns to the free list: later tensors of the plan may reuse it (one in-order stream)
        return out

    def _gn_stats(self, n, hw, c):
        # per image: the per-channel affine [C][2], then partial sums [blocks][32][2] for blocks of 64 pixels (the strip
        # convolution's granularity; 128-pixel producers use the first half)
        return self.alloc((n * (2 * c + 64 * ((hw + 63) // 64)), ), f32)

    def gn(self, x, holder, swish=True):
        n, h, wd, c = x.shape
        if self.split:
            out = self.alloc((2, n, h, wd, c), bf16)
            out.is_planes, out.shape = True, (n, h, wd, c)
            st = getattr(x, 'gn_stats', None)  # partial sums already written by the producing convolution's epilogue
            flags = self.SPLIT | (2 if st is not None else 0) | (8 if getattr(st, 'blocks64', False) else 0)
            if st is None:
                st = self._gn_stats(n, h * wd, c)
            i = self._op(op=self.OP_GN, mode=int(swish), N=n, H=h, W=wd, C=c, flags=flags, in0=x.off, out_bf16=out.off,
                         scratch=st.off, w=holder.weight.data_ptr(), b=holder.bias.data_ptr(), eps=1e-6)
            out.gn_op = self.ops[i]  # (see _conv_split: its reader may switch it to the fp16 output)
            return out
        if self.strict:
            out = self.alloc(x.shape, f32)
            st = self.alloc((n * 2 * c, ), f32)
            self._op(op=self.OP_GN, mode=int(swish), N=n, H=h, W=wd, C=c, flags=self.STRICT, in0=x.off, out_f32=out.off,
                     scratch=st.off, w=holder.weight.data_ptr(), b=holder.bias.data_ptr(), eps=1e-6)
            return out
        out = self.alloc(x.shape, bf16)
        st = getattr(x, 'gn_stats', None)
        flags = (1 if x.dtype == f32 else 0) | (2 if st is not None else 0) | (8 if getattr(st, 'blocks64', False) else 0)
        if st is None:
            st = self._gn_stats(n, h * wd, c)
        self._op(op=self.OP_GN, mode=int(swish), N=n, H=h, W=wd, C=c, flags=flags, in0=x.off,
                 out_bf16=out.off, scratch=st.off, w=holder.weight.data_ptr(), b=holder.bias.data_ptr(), eps=1e-6)
        return out

    def cast(self, x):
        if self.split:
            return self._planes(x)
        if x.dtype == bf16 or self.strict:
            return x
        if getattr(x, 'bf16', None) is not None:  # the producing conv already stored the bf16 copy
            return x.bf16
        out = self.alloc(x.shape, bf16)
        n, h, wd, c = x.shape
        self._op(op=self.OP_CAST, N=n, H=h, W=wd, C=c, in0=x.off, out_bf16=out.off)
        return out

    def conv_qkv(self, x, blk):
        """AttnBlock q / k / v (model.py:159-178: three 1x1 convs of the same input) as ONE 1x1 conv with the three weights
        stacked along Cout: one launch instead of three small ones.  Returns the [n, h, w, 3c] buffer (q, k, v are its column blocks)
        and c."""
        w, b = self.vae._cw_qkv(blk)
        n, h, wd, cin = x.shape
        c = w.shape[0] // 3
        out = self.alloc((n, h, wd, 3 * c), bf16)
        self._op(op=self.OP_CONV, mode=3, N=n, H=h, W=wd, C=cin, Cout=3 * c, flags=0, in0=x.off, in1=-1, out_bf16=out.off, out_f32=-1,
                 scratch=-1, w=w.data_ptr(), b=b.data_ptr())
        return out, c

    def spatial_attention_fused(self, qkv, c):
        n, h, wd, c3 = qkv.shape
        hw = h * wd
        out = self.alloc((n, h, wd, c), bf16)
        sc = self.alloc((n * hw * hw * 3 // 2 + 64, ), f32)
        self._op(op=self.OP_ATTN, N=n, H=h, W=wd, C=c, in0=qkv.off, in1=qkv.off + 2 * c, in2=qkv.off + 4 * c, out_bf16=out.off,
                 scratch=sc.off, eps=float(c)**-0.5, pad=c3)
        return out

    def spatial_attention(self, q, k, v):
        n, h, wd, c = q.shape
        hw = h * wd
        if self.strict or self.split:  # (split: q, k, v are fp32 conv outputs; the fp32 attention is 0.2 % of the encoder's work)
            out = self.alloc(q.shape, f32)
            sc = self.alloc((2 * n * hw * hw, ), f32)
            self._op(op=self.OP_ATTN, N=n, H=h, W=wd, C=c, flags=self.STRICT, in0=q.off, in1=k.off, in2=v.off,
                     out_f32=out.off, scratch=sc.off, eps=float(c)**-0.5)
            return out
        out = self.alloc(q.shape, bf16)
        sc = self.alloc((n * hw * hw * 3 // 2 + 64, ), f32)
        self._op(op=self.OP_ATTN, N=n, H=h, W=wd, C=c, in0=q.off, in1=k.off, in2=v.off, out_bf16=out.off,
                 scratch=sc.off, eps=float(c)**-0.5)
        return out

    def vq_argmin(self, z):
        n, h, wd, c = z.shape
        cb = self.vae.model.quantize.embedding.weight
        i = self._op(op=self.OP_VQ, N=n, H=h, W=wd, C=c, Cout=cb.shape[0], in0=z.off, w=cb.data_ptr(),
                     b=self.vae._ee().data_ptr())
        self.patches.append((i, 'ext_out', 'idx'))

    def gather(self, n, hw):
        cb = self.vae.model.quantize.embedding.weight
        f32out = self.strict or self.split
        out = self.alloc((n, hw, hw, cb.shape[1]), f32 if f32out else bf16)
        i = self._op(op=self.OP_GATHER, N=n, H=hw, W=hw, C=cb.shape[1], Cout=cb.shape[0], w=cb.data_ptr(),
                     flags=self.STRICT if f32out else 0, **{'out_f32' if f32out else 'out_bf16': out.off})
        self.patches.append((i, 'ext_in', 'idx'))
        return out

    def external_z(self, n, hw, c):
        """decode_train: z [n*hw*hw, c] fp32 computed outside the plan (probs @ codebook) enters here."""
        f32out = self.strict or self.split
        out = self.alloc((n, hw, hw, c), f32 if f32out else bf16)
        i = self._op(op=self.OP_EXT, N=n, H=hw, W=hw, C=c, flags=self.STRICT if f32out else 0,
                     **{'out_f32' if f32out else 'out_bf16': out.off})
        self.patches.append((i, 'ext_in', 'z'))
        return out

    def to_nchw(self, x, cuse):
        n, h, wd, c = x.shape
        i = self._op(op=self.OP_NCHW, N=n, H=h, W=wd, C=c, Cout=cuse, in0=x.off)
        self.patches.append((i, 'ext_out', 'img'))

    def keep(self, name, buf):
        """Pin a planned tensor so it can be read back after run() (tests / encode_z)."""
        self.kept[name] = (buf.off, buf.shape)
        self._pinned = getattr(self, '_pinned', []) + [buf]

    def finish(self, device):
        self.recording = False
        arr = (_lib.VqganOp * len(self.ops))(*self.ops)
        arena = torch.empty(max(self.top, 256), device=device, dtype=torch.uint8)
        return _Plan(arr, arena, self.patches, self.kept)


class VQGanVAE1024(nn.Module):
    def __init__(self, vae_path=None, image_size=None, ddconfig=None, n_embed=1024, embed_dim=256):
        super().__init__()
        cfg = dict(DEFAULT_DDCONFIG)
        cfg.update(ddconfig or {})
        if image_size:
            cfg['resolution'] = image_size  # vae.py:24-25
        self.model = VQModel(cfg, n_embed, embed_dim)
        if vae_path is not None:
            state = torch.load(vae_path, map_location='cpu')['state_dict']  # vae.py:28-30
            self.model.load_state_dict(state, strict=False)
        self.num_layers = 4
        self.image_size = 256
        self.num_tokens = 1024
        # Arithmetic modes of the encoder / decoder, and what each means for the token indices (measured: the flip census of 40,960 fresh
        # full-size tokens per mode, profiles/r06_flip_census.log, and 2 x 1,024 reference tokens, tests/test_round6_gpu.py):
        #   strict = False   bf16 MFMA operands (training speed): 97.7 % of the reference's tokens; every flip a near-tie of its distances
        #   strict = 'mixed' the ENCODER's 3x3 residual-block convolutions on maps of at least mixed_f16_side pixels a side (the 128x128,
        #                    64x64 and 32x32 levels: 82 % of the multiply-adds) as ONE product of fp16 operands, the rest as 'split':
        #                    99.87 % of the tokens at 1.11x the step -- NOT an exact mode (rounds 4-5 called it one on 448 golden tokens)
        #   strict = 'split' bf16-pair convolutions (3 products per convolution, fp32 accumulate; ~1e-5 of the fp32 result), fp32 residual
        #                    stream / GroupNorm / attention: the reference's tokens except ties at its OWN fp32 resolution (2 in 40,960
        #                    against the fp32 mode; the one golden flip has a top-2 gap of 17 fp32 spacings of the distance); 1.28x
        #   strict = True    fp32 operator (csrc/strict.hip: f32 MFMA, fp64 GroupNorm statistics): every token seen equal; 2.3x
        # (per-layer sensitivity behind 'mixed': tests/sweep_exact_layers.py, profiles/r05_exact_index_layer_sensitivity_sweep.log)
        self.strict = False
        self.mixed_f16_side = 32  # (the error comes from the 128x128 level: 64 -> 32 adds 3 % to max |dz|, the sweep's rows A / B)
        # default (bf16) operator only: 'bf16' = the ENCODER's residual stream between blocks is bf16 too (round 5: the fp32 stream
        # cost 0.3 ms of the training step in stores / GroupNorm reads and bought nothing the default mode promises -- its indices
        # are 97-100 % of the reference's either way); 'bf16_all' = the decoder's as well; 'f32' = fp32
        # residual streams (rounds 1-4)
        self.stream = os.environ.get('MMVID_VQGAN_STREAM', 'bf16')
        self._prep = {}
        self._prep_key = None

    # ---- weight preparation (cached) ----------------------------------------------------------------
    def _prepared(self):
        key = tuple((p._version, p.data_ptr()) for p in self.model.parameters())
        if key != self._prep_key:
            self._prep = {}
            self._prep_key = key
        return self._prep

    def _cw(self, holder, strict=False):
        """conv holder -> (w bf16 [Cout_p, taps, Cin_p], bias f32 [Cout_p], Cout); strict: w fp32 [Cout, taps, Cin_p4]."""
        prep = self._prepared()
        k = (id(holder), strict)
        if strict and k not in prep:
            w, b = holder.weight.detach().float(), holder.bias.detach().float()
            cout, cin, kh, kw = w.shape
            cin_p = max(4, _pow2_at_least8(cin) if cin > 4 else 4)
            wp = torch.zeros(cout, kh * kw, cin_p, device=w.device, dtype=f32)
            wp[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
            prep[k] = (wp.contiguous(), b.contiguous(), cout)
        if k not in prep:
            w, b = holder.weight.detach(), holder.bias.detach()
            cout, cin, kh, kw = w.shape
            cin_p, cout_p = _pow2_at_least8(cin), (cout + 7) // 8 * 8
            wp = torch.zeros(cout_p, kh * kw, cin_p, device=w.device, dtype=f32)
            wp[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
            bp = torch.zeros(cout_p, device=w.device, dtype=f32)
            bp[:cout] = b
            prep[k] = (wp.to(bf16).contiguous(), bp, cout)
        return prep[k]

    def _cw_split(self, holder):
        """conv holder -> (w3 bf16 [Cout_p, 3, taps, Cin_p] = (w_hi | w_hi | w_lo), bias f32 [Cout_p], Cout): the weight side of
        the split operator, w = w_hi + w_lo with w_hi = bf16(w), w_lo = bf16(w - w_hi)."""
        prep = self._prepared()
        k = (id(holder), 'split')
        if k not in prep:
            w, b = holder.weight.detach().float(), holder.bias.detach().float()
            cout, cin, kh, kw = w.shape
            cin_p, cout_p = _pow2_at_least8(cin), (cout + 7) // 8 * 8
            wp = torch.zeros(cout_p, kh * kw, cin_p, device=w.device, dtype=f32)
            wp[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
            hi = wp.to(bf16)
            lo = (wp - hi.float()).to(bf16)
            bp = torch.zeros(cout_p, device=w.device, dtype=f32)
            bp[:cout] = b
            prep[k] = (torch.stack([hi, hi, lo], 1).contiguous(), bp, cout)
        return prep[k]

    def _cw_f16(self, holder):
        """conv holder -> (w fp16 [Cout_p, taps, Cin_p], bias f32 [Cout_p], Cout): the weight side of the fp16 single-product form."""
        prep = self._prepared()
        k = (id(holder), 'f16')
        if k not in prep:
            w, b = holder.weight.detach().float(), holder.bias.detach().float()
            cout, cin, kh, kw = w.shape
            cin_p, cout_p = _pow2_at_least8(cin), (cout + 7) // 8 * 8
            wp = torch.zeros(cout_p, kh * kw, cin_p, device=w.device, dtype=f32)
            wp[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
            bp = torch.zeros(cout_p, device=w.device, dtype=f32)
            bp[:cout] = b
            prep[k] = (wp.to(torch.float16).contiguous(), bp, cout)
        return prep[k]

    def _cw_qkv(self, blk):
        """q, k, v 1x1 conv holders of an AttnBlock -> (w bf16 [3C, 1, C], bias f32 [3C])."""
        prep = self._prepared()
        key = (id(blk), 'qkv')
        if key not in prep:
            ws, bs = zip(*[self._cw(hd)[:2] for hd in (blk.q, blk.k, blk.v)])
            prep[key] = (torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous())
        return prep[key]

    def _ee(self):
        prep = self._prepared()
        if 'ee' not in prep:
            prep['ee'] = ops.vq_sqnorm(self.model.quantize.embedding.weight.detach().contiguous())
        return prep['ee']

    # ---- planning: the op sequence of one encode / decode for a given batch shape ------------------------------
    def _plan(self, kind, n, size_or_hw, slot=0):
        prep = self._prepared()
        mode = 'split' if self.strict in ('split', 'mixed') else bool(self.strict)
        # (the decoder keeps its fp32 stream unless 'bf16_all': it is off the training path and its pixel tolerance is pinned)
        s16 = mode is False and (self.stream == 'bf16_all' or (self.stream == 'bf16' and kind == 'enc'))
        f16_side = self.mixed_f16_side if (self.strict == 'mixed' and kind == 'enc') else 0  # (the decoder stays the pair operator)
        key = ('plan', kind, n, size_or_hw, mode, slot, s16, f16_side)  # (slot: plans that run concurrently need arenas of their own)
        if key not in prep:
            pl = _Planner(self, strict=mode, stream16=s16, f16_side=f16_side)
            if kind == 'enc':
                self._plan_encode(pl, n, size_or_hw)
            elif kind == 'dec_z':
                self._plan_decode(pl, n, size_or_hw, from_z=True)
            else:
                self._plan_decode(pl, n, size_or_hw)
            prep[key] = pl.finish(next(self.model.parameters()).device)
        return prep[key]

    def _plan_resblock(self, pl, x32, blk, final='f32'):
        """model.py:130-150 on an fp32 residual stream.  final: 'f32' (residual stream continues), 'both' (a conv reads
        the result next as well) or 'bf16' (ONLY a conv reads it: no fp32 store at all)."""
        h = pl.conv(pl.gn(x32, blk.norm1), blk.conv1, 0, feeds_gn=True)
        h = pl.gn(h, blk.norm2)
        skip = x32
        if hasattr(blk, 'nin_shortcut'):
            skip = pl.conv(pl.cast(x32), blk.nin_shortcut, 3, out32=True)
        return pl.conv(h, blk.conv2, 0, residual=skip, out32=final != 'bf16', feeds_gn=final != 'bf16',
                       also_bf16=final == 'both')

    def _plan_attn(self, pl, x32, blk, final='f32'):
        """model.py:180-205."""
        h = pl.gn(x32, blk.norm, swish=False)
        if _FUSE_QKV and not pl.strict and not pl.split:
            qkv, c = pl.conv_qkv(h, blk)
            o = pl.spatial_attention_fused(qkv, c)
        else:
            q, k, v = pl.conv(h, blk.q, 3), pl.conv(h, blk.k, 3), pl.conv(h, blk.v, 3)
            o = pl.spatial_attention(q, k, v)
        return pl.conv(o, blk.proj_out, 3, residual=x32, out32=final != 'bf16', feeds_gn=final != 'bf16',
                       also_bf16=final == 'both')

    def _plan_encode(self, pl, n, s):
        """Encoder.forward (model.py:439-466) + quant_conv (vqgan.py:67-68) + VQ lookup (quantize.py:302-310)."""
        enc = self.model.encoder

        def needs_bf16(blk):  # a resblock whose shortcut is a 1x1 conv reads its input in bf16 too
            return hasattr(blk, 'nin_shortcut')

        h = pl.conv(pl.image(n, s), enc.conv_in, 0, out32=True, feeds_gn=True, also_bf16=needs_bf16(enc.down[0].block[0]))
        for li, d in enumerate(enc.down):
            has_down = hasattr(d, 'downsample')
            nxt = enc.down[li + 1].block[0] if li + 1 < len(enc.down) else enc.mid.block_1
            for bi, blk in enumerate(d.block):
                last = bi == len(d.block) - 1
                with_attn = len(d.attn) > 0
                # what the level's last tensor feeds: only the downsample conv (bf16) / the next block's shortcut too
                end = 'bf16' if has_down else ('both' if needs_bf16(nxt) else 'f32')
                mid = 'both' if (not last and needs_bf16(d.block[bi + 1])) else 'f32'
                want = (end if last else mid) if _DUAL_OUT else 'f32'
                h = self._plan_resblock(pl, h, blk, final='f32' if with_attn else want)
                if with_attn:
                    h = self._plan_attn(pl, h, d.attn[bi], final=want)
            if has_down:
                h = pl.conv(pl.cast(h), d.downsample.conv, 1, out32=True, feeds_gn=True, also_bf16=needs_bf16(nxt))
        h = self._plan_resblock(pl, h, enc.mid.block_1)
        h = self._plan_attn(pl, h, enc.mid.attn_1)
        h = self._plan_resblock(pl, h, enc.mid.block_2)
        h = pl.conv(pl.gn(h, enc.norm_out), enc.conv_out, 0)
        z = pl.conv(h, self.model.quant_conv, 3, out32=True, keep32=True)  # [N, h, w, embed_dim] fp32 = VQ rows
        pl.keep('z', z)
        pl.vq_argmin(z)

    def _plan_decode(self, pl, n, hw, from_z=False):
        """codebook gather (vae.py:50) + post_quant_conv + Decoder.forward (model.py:551-582) + vae.py:55.
        from_z: the quantised map arrives as an external fp32 tensor instead (decode_train, vae.py:58-68)."""
        dec = self.model.decoder
        z0 = pl.external_z(n, hw, self.model.quantize.embedding.weight.shape[1]) if from_z else pl.gather(n, hw)
        h = pl.conv(z0, self.model.post_quant_conv, 3)
        h = pl.conv(h, dec.conv_in, 0, out32=True, feeds_gn=True)
        h = self._plan_resblock(pl, h, dec.mid.block_1)
        h = self._plan_attn(pl, h, dec.mid.attn_1)
        h = self._plan_resblock(pl, h, dec.mid.block_2)
        for lvl in reversed(range(len(dec.up))):
            u = dec.up[lvl]
            for bi, blk in enumerate(u.block):
                h = self._plan_resblock(pl, h, blk)
                if len(u.attn) > 0:
                    h = self._plan_attn(pl, h, u.attn[bi])
            if hasattr(u, 'upsample'):
                h = pl.conv(pl.cast(h), u.upsample.conv, 2, out32=True, feeds_gn=True)
        h = pl.gn(h, dec.norm_out)
        img = pl.conv(h, dec.conv_out, 0, out32=True, clamp01=True, keep32=True)  # (clamp(x,-1,1)+1)/2 fused, vae.py:55
        pl.to_nchw(img, 3)

    # ---- reference API ------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_z(self, img):
        """img [N,3,S,S] fp32 in [0,1] -> pre-quantisation z [N, h, w, embed_dim] fp32 (NHWC; a copy)."""
        step, outs = self._max_frames(img.shape[-1]), []
        for i in range(0, max(img.shape[0], 1), step):  # (z lives in the plan's arena: read it back slice by slice)
            _, plan = self._encode(img[i:i + step])
            off, shape = plan.kept['z']
            nbytes = int(torch.tensor(shape).prod()) * 4
            outs.append(plan.arena[off:off + nbytes].view(torch.float32).view(shape).clone())
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    def _max_frames(self, s):
        """Frames per planned call: the kernels address an operand through 32-bit byte offsets (< 2 GiB per tensor), and the largest operand
        of a plan is a pair of bf16 planes at the full resolution with `ch` channels."""
        ch = self.model.ddconfig['ch']
        return max(1, ((1 << 31) - 1) // (s * s * max(ch, 8) * 4))

    def _encode(self, img):
        img = ops._chk(img.contiguous().float(), f32, 'img')
        n, c, s, s2 = img.shape
        assert c == 3 and s == s2
        idx = torch.empty(n, (s // 16)**2, device=img.device, dtype=torch.int64)
        step = self._max_frames(s)
        plan = None
        for i in range(0, max(n, 1), step):  # (one call for every batch the drivers use; more than 255 full-size frames go in slices)
            m = min(step, n - i)
            plan = self._plan('enc', m, s)
            plan.run(ext_in={'img': img[i:i + m]}, ext_out={'idx': idx[i:i + m]})
        return idx, plan

    @torch.no_grad()
    def get_codebook_indices(self, img):
        """vae.py:38-43: [N,3,S,S] in [0,1] -> [N, (S/16)^2] int64."""
        return self._encode(img)[0]

    def decode(self, img_seq):
        """vae.py:45-56: [N, n] int64 -> [N,3,S,S] fp32 in [0,1]."""
        with torch.no_grad():
            img_seq = ops._chk(img_seq.contiguous(), torch.int64, 'img_seq')
            b, n = img_seq.shape
            hw = int(sqrt(n))
            out = torch.empty(b, 3, hw * 16, hw * 16, device=img_seq.device, dtype=f32)
            step = self._max_frames(hw * 16)
            for i in range(0, b, step):  # (slices of at most 255 full-size frames: 32-bit operand offsets)
                m = min(step, b - i)
                self._plan('dec', m, hw).run(ext_in={'idx': img_seq[i:i + m]}, ext_out={'img': out[i:i + m]})
            return out

    def decode_train(self, probs):
        """vae.py:58-68: probs [B, N, num_tokens] (soft one-hot) @ codebook -> decoder -> [B,3,S,S] in [0,1].
        The soft lookup is the exact-fp32 matrix kernel; the decoder is the planned op list of `decode`.  The VQGAN is
        frozen and no caller of the reference differentiates through it, so this is a forward-only entry point."""
        with torch.no_grad():
            probs = ops._chk(probs.contiguous().float(), f32, 'probs')
            b, n, d = probs.shape
            cb = self.model.quantize.embedding.weight.detach()
            assert d == cb.shape[0], f'probs last dim {d} != codebook size {cb.shape[0]}'
            hw = int(sqrt(n))
            z = ops.gemm_f32(probs.view(b * n, d), cb, b_kmajor=True)  # [b*n, embed_dim] = probs @ codebook
            plan = self._plan('dec_z', b, hw)
            out = torch.empty(b, 3, hw * 16, hw * 16, device=probs.device, dtype=f32)
            plan.run(ext_in={'z': z}, ext_out={'img': out})
            return out

    def forward(self, img):
        raise NotImplementedError  # as the reference (vae.py:70-71)
