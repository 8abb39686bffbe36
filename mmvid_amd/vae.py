"""Host-side mirror of mmvid_pytorch/vae.py::VQGanVAE1024 (15-71) -> taming VQModel (vqgan.py:16-75) over the HIP
kernels: same constructor, attributes (`image_size` is mutated by the driver, train.py:183; `num_layers`,
`num_tokens`), methods (`get_codebook_indices`, `decode`, `decode_train`) and the same state_dict keys as
`model.encoder.* / model.decoder.* / model.quantize.embedding.weight / model.quant_conv.* / model.post_quant_conv.*`
(taming/modules/diffusionmodules/model.py:363-582), so published `vqgan.1024.model.ckpt` files load.

Execution (csrc/conv.hip, norm.hip, vq.hip): activations are NHWC; the residual stream is fp32, every conv
input is bf16 (MFMA, fp32 accumulate); GroupNorm statistics are fp32.  The frozen weights are re-laid-out once
into [Cout][ky][kx][Cin] bf16 (cached, refreshed when a parameter changes)."""
from math import sqrt

import torch
from torch import nn

from . import ops

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


def _pow2_at_least8(c):
    p = 8
    while p < c:
        p *= 2
    return p


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
        self._prep = {}
        self._prep_key = None

    # ---- weight preparation (cached) ----------------------------------------------------------------
    def _prepared(self):
        key = tuple((p._version, p.data_ptr()) for p in self.model.parameters())
        if key != self._prep_key:
            self._prep = {}
            self._prep_key = key
        return self._prep

    def _cw(self, holder):
        """conv holder -> (w bf16 [Cout_p, taps, Cin_p], bias f32 [Cout_p], Cout)."""
        prep = self._prepared()
        k = id(holder)
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

    def _ee(self):
        prep = self._prepared()
        if 'ee' not in prep:
            prep['ee'] = ops.vq_sqnorm(self.model.quantize.embedding.weight.detach().contiguous())
        return prep['ee']

    # ---- building blocks ----------------------------------------------------------------------------
    def _conv(self, x16, holder, mode, residual=None, out32=False, clamp01=False):
        w, b, _ = self._cw(holder)
        return ops.conv2d_nhwc(x16, w, b, mode, residual=residual, clamp01=clamp01, out_dtype=f32 if out32 else bf16)

    def _gn(self, x, holder, swish=True):
        return ops.groupnorm_swish(x, holder.weight.detach(), holder.bias.detach(), 1e-6, swish, bf16)

    def _resblock(self, x32, blk):
        """model.py:130-150 on an fp32 residual stream."""
        h = self._conv(self._gn(x32, blk.norm1), blk.conv1, 0)
        h = self._gn(h, blk.norm2)
        skip = x32
        if hasattr(blk, 'nin_shortcut'):
            skip = self._conv(ops.cast_bf16(x32), blk.nin_shortcut, 3, out32=True)
        return self._conv(h, blk.conv2, 0, residual=skip, out32=True)

    def _attn(self, x32, blk):
        """model.py:180-205."""
        h = self._gn(x32, blk.norm, swish=False)
        n, hh, ww, c = h.shape
        q, k, v = (self._conv(h, m, 3).view(n, hh * ww, c) for m in (blk.q, blk.k, blk.v))
        o = ops.spatial_attention(q, k, v).view(n, hh, ww, c)
        return self._conv(o, blk.proj_out, 3, residual=x32, out32=True)

    # ---- reference API ------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_z(self, img):
        """img [N,3,S,S] fp32 in [0,1] -> pre-quantisation z rows [N*hw, embed_dim] fp32 (NHWC order)."""
        enc = self.model.encoder
        h = self._conv(ops.image_to_nhwc8(img.contiguous().float()), enc.conv_in, 0, out32=True)
        for d in enc.down:
            for bi, blk in enumerate(d.block):
                h = self._resblock(h, blk)
                if len(d.attn) > 0:
                    h = self._attn(h, d.attn[bi])
            if hasattr(d, 'downsample'):
                h = self._conv(ops.cast_bf16(h), d.downsample.conv, 1, out32=True)
        h = self._resblock(h, enc.mid.block_1)
        h = self._attn(h, enc.mid.attn_1)
        h = self._resblock(h, enc.mid.block_2)
        h = self._conv(self._gn(h, enc.norm_out), enc.conv_out, 0)
        z = self._conv(h, self.model.quant_conv, 3, out32=True)  # [N, h, w, embed_dim]
        return z

    @torch.no_grad()
    def get_codebook_indices(self, img):
        """vae.py:38-43: [N,3,S,S] in [0,1] -> [N, (S/16)^2] int64."""
        b = img.shape[0]
        z = self.encode_z(img)
        cb = self.model.quantize.embedding.weight.detach().contiguous()
        idx = ops.vq_argmin(z.view(-1, z.shape[-1]), cb, self._ee())
        return idx.view(b, -1)

    def decode(self, img_seq):
        """vae.py:45-56: [N, n] int64 -> [N,3,S,S] fp32 in [0,1]."""
        with torch.no_grad():
            b, n = img_seq.shape
            hw = int(sqrt(n))
            cb = self.model.quantize.embedding.weight.detach().contiguous()
            z = ops.gather_rows(cb, img_seq.contiguous(), bf16).view(b, hw, hw, -1)
            return self._decode_z(z)

    def decode_train(self, probs):
        """vae.py:58-68: probs [B, N, n_embed] (soft one-hot) -> images; no gradient through this frozen path."""
        with torch.no_grad():
            b, n, d = probs.shape
            hw = int(sqrt(n))
            cbT = ops.cast_bf16(self.model.quantize.embedding.weight.detach().t().contiguous())  # [256, n_embed]
            z = ops.gemm(ops.cast_bf16(probs.reshape(b * n, d).contiguous().float()), cbT).view(b, hw, hw, -1)
            return self._decode_z(z)

    def _decode_z(self, z16):
        dec = self.model.decoder
        h = self._conv(z16.contiguous(), self.model.post_quant_conv, 3)
        h = self._conv(h, dec.conv_in, 0, out32=True)
        h = self._resblock(h, dec.mid.block_1)
        h = self._attn(h, dec.mid.attn_1)
        h = self._resblock(h, dec.mid.block_2)
        for lvl in reversed(range(len(dec.up))):
            u = dec.up[lvl]
            for bi, blk in enumerate(u.block):
                h = self._resblock(h, blk)
                if len(u.attn) > 0:
                    h = self._attn(h, u.attn[bi])
            if hasattr(u, 'upsample'):
                h = self._conv(ops.cast_bf16(h), u.upsample.conv, 2, out32=True)
        h = self._gn(h, dec.norm_out)
        img = self._conv(h, dec.conv_out, 0, out32=True, clamp01=True)  # (clamp(x,-1,1)+1)/2 fused, vae.py:55
        return ops.nhwc_to_nchw(img, 3)

    def forward(self, img):
        raise NotImplementedError  # as the reference (vae.py:70-71)
