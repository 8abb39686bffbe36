"""VQGAN encode / quantise / decode oracle (fp32 torch-CPU, functional over a state_dict).

TEST INFRASTRUCTURE.  Restates, without nn.Modules:
  taming/modules/diffusionmodules/model.py  Encoder 363-466, Decoder 469-582, ResnetBlock 87-150,
      AttnBlock 153-205, Downsample 65-84, Upsample 45-62, Normalize 38-42, nonlinearity 33-35
  taming/models/vqgan.py  VQModel.encode 66-70 / decode 72-75
  mmvid_pytorch/vae.py    get_codebook_indices 38-43 / decode 45-56
Config (mmvid_pytorch/data/vqgan.1024.config.yml): ch_mult (1,1,2,2,4), 2 res blocks, attention
where the CURRENT resolution == 16, GroupNorm(32, eps 1e-6), swish, z_channels = embed_dim = 256.
"""
import torch
import torch.nn.functional as F

from .vq import vq_argmin

CH_MULT = (1, 1, 2, 2, 4)
NUM_RES_BLOCKS = 2
ATTN_RES = (16, )


def _gn_swish(sd, p, x):
    h = F.group_norm(x, 32, sd[p + '.weight'], sd[p + '.bias'], eps=1e-6)
    return h * torch.sigmoid(h)


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + '.weight'], sd[p + '.bias'], eps=1e-6)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=stride, padding=padding)


def resnet_block(sd, p, x):
    """model.py:130-150 (temb None, dropout 0)."""
    h = _conv(sd, p + '.conv1', _gn_swish(sd, p + '.norm1', x))
    h = _conv(sd, p + '.conv2', _gn_swish(sd, p + '.norm2', h))
    if (p + '.nin_shortcut.weight') in sd:
        x = _conv(sd, p + '.nin_shortcut', x, padding=0)
    return x + h


def attn_block(sd, p, x):
    """model.py:180-205: single-head softmax(q k^T c^-0.5) v over the HW positions."""
    h = _gn(sd, p + '.norm', x)
    q = _conv(sd, p + '.q', h, padding=0)
    k = _conv(sd, p + '.k', h, padding=0)
    v = _conv(sd, p + '.v', h, padding=0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.softmax(torch.bmm(q, k) * (int(c)**(-0.5)), dim=2)
    v = v.reshape(b, c, hh * ww)
    o = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + '.proj_out', o, padding=0)


def downsample(sd, p, x):
    """model.py:77-81: zero pad right/bottom by one, conv3x3 stride 2 no padding."""
    return _conv(sd, p + '.conv', F.pad(x, (0, 1, 0, 1)), stride=2, padding=0)


def upsample(sd, p, x):
    """model.py:56-62: nearest x2 then conv3x3."""
    return _conv(sd, p + '.conv', F.interpolate(x, scale_factor=2.0, mode='nearest'))


def encoder(sd, x, resolution, prefix='encoder'):
    p = prefix
    cur = resolution
    h = _conv(sd, p + '.conv_in', x)
    for lvl in range(len(CH_MULT)):
        for b in range(NUM_RES_BLOCKS):
            h = resnet_block(sd, f'{p}.down.{lvl}.block.{b}', h)
            if cur in ATTN_RES:
                h = attn_block(sd, f'{p}.down.{lvl}.attn.{b}', h)
        if lvl != len(CH_MULT) - 1:
            h = downsample(sd, f'{p}.down.{lvl}.downsample', h)
            cur //= 2
    h = resnet_block(sd, p + '.mid.block_1', h)
    h = attn_block(sd, p + '.mid.attn_1', h)
    h = resnet_block(sd, p + '.mid.block_2', h)
    return _conv(sd, p + '.conv_out', _gn_swish(sd, p + '.norm_out', h))


def decoder(sd, z, resolution, prefix='decoder'):
    p = prefix
    n = len(CH_MULT)
    cur = resolution // 2**(n - 1)
    h = _conv(sd, p + '.conv_in', z)
    h = resnet_block(sd, p + '.mid.block_1', h)
    h = attn_block(sd, p + '.mid.attn_1', h)
    h = resnet_block(sd, p + '.mid.block_2', h)
    for lvl in reversed(range(n)):
        for b in range(NUM_RES_BLOCKS + 1):
            h = resnet_block(sd, f'{p}.up.{lvl}.block.{b}', h)
            if cur in ATTN_RES:
                h = attn_block(sd, f'{p}.up.{lvl}.attn.{b}', h)
        if lvl != 0:
            h = upsample(sd, f'{p}.up.{lvl}.upsample', h)
            cur *= 2
    return _conv(sd, p + '.conv_out', _gn_swish(sd, p + '.norm_out', h))


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def encode_z(sd, img, image_size, prefix='model.'):
    """img [N,3,S,S] in [0,1] -> pre-quantisation z_e [N,256,S/16,S/16] (vae.py:41, vqgan.py:67-68)."""
    m = _sub(sd, prefix)
    h = encoder(m, 2 * img - 1, image_size)
    return _conv(m, 'quant_conv', h, padding=0)


def get_codebook_indices(sd, img, image_size, prefix='model.'):
    """vae.py:38-43 -> [N, (S/16)^2] int64."""
    z = encode_z(sd, img, image_size, prefix)
    n, c = z.shape[:2]
    zf = z.permute(0, 2, 3, 1).reshape(-1, c)  # quantize.py:302-303
    idx, _ = vq_argmin(zf, sd[prefix + 'quantize.embedding.weight'])
    return idx.view(n, -1)


def decode(sd, img_seq, image_size, prefix='model.'):
    """vae.py:45-56: codebook gather -> NCHW -> post_quant_conv -> Decoder -> (clamp+1)/2."""
    m = _sub(sd, prefix)
    b, n = img_seq.shape
    hw = int(round(n**0.5))
    z = m['quantize.embedding.weight'][img_seq]  # b n c
    z = z.view(b, hw, hw, -1).permute(0, 3, 1, 2)
    h = _conv(m, 'post_quant_conv', z, padding=0)
    img = decoder(m, h, image_size)
    return (img.clamp(-1., 1.) + 1) * 0.5
