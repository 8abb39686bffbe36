"""CLIP ViT-B/32-shaped transformer tower oracle (fp32 torch-CPU, functional).

TEST INFRASTRUCTURE.  Restates mmvid_pytorch/transformers/clip_model.py:
  ResidualAttentionBlock 201-227 (x += MHA(LN1 x); x += c_proj(QuickGELU(c_fc(LN2 x)))),
  LayerNorm 188-193 (eps 1e-5), QuickGELU 196-198, nn.MultiheadAttention packed in_proj
  (rows 0:E = Q, E:2E = K, 2E:3E = V; heads = contiguous 64-wide slices; scale hd^-0.5),
  OpenAICLIPTransformer.build_attention_mask 561-578 and forward 580-584.
Batch-first [B, L, E] here; the reference permutes to [L, B, E] internally (same maths).
"""
import math

import torch
import torch.nn.functional as F


def build_attention_mask(L, mask_type='causal', index=()):
    """clip_model.py:561-578: additive float mask [L, L]."""
    if mask_type == 'causal':
        return torch.full((L, L), float('-inf')).triu_(1)
    if mask_type == 'mask_prev':
        m = torch.zeros(L, L)
        for i in index:
            m[i, :i] = float('-inf')
        return m
    raise NotImplementedError(mask_type)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def mha(sd, p, x, mask, heads):
    B, L, E = x.shape
    hd = E // heads
    qkv = F.linear(x, sd[p + '.in_proj_weight'], sd[p + '.in_proj_bias'])
    q, k, v = qkv.split(E, dim=-1)

    def sp(t):
        return t.view(B, L, heads, hd).transpose(1, 2)  # B H L hd

    s = torch.matmul(sp(q), sp(k).transpose(-1, -2)) / math.sqrt(hd)
    if mask is not None:
        s = s + mask[:L, :L]
    o = torch.matmul(torch.softmax(s, dim=-1), sp(v))
    o = o.transpose(1, 2).reshape(B, L, E)
    return F.linear(o, sd[p + '.out_proj.weight'], sd[p + '.out_proj.bias'])


def res_block(sd, p, x, mask, heads):
    E = x.shape[-1]
    h = F.layer_norm(x, (E, ), sd[p + '.ln_1.weight'], sd[p + '.ln_1.bias'], 1e-5)
    x = x + mha(sd, p + '.attn', h, mask, heads)
    h = F.layer_norm(x, (E, ), sd[p + '.ln_2.weight'], sd[p + '.ln_2.bias'], 1e-5)
    h = quick_gelu(F.linear(h, sd[p + '.mlp.c_fc.weight'], sd[p + '.mlp.c_fc.bias']))
    return x + F.linear(h, sd[p + '.mlp.c_proj.weight'], sd[p + '.mlp.c_proj.bias'])


def num_layers(sd, prefix):
    return len({k[len(prefix):].split('.')[1] for k in sd if k.startswith(prefix + 'resblocks.')})


def tower(sd, x, mask, prefix='transformer.', heads=None):
    """x [B, L, E] -> [B, L, E]; prefix e.g. 'transformer.' (keys '<prefix>resblocks.<i>...')."""
    E = x.shape[-1]
    heads = heads or E // 64
    for i in range(num_layers(sd, prefix)):
        x = res_block(sd, f'{prefix}resblocks.{i}', x, mask, heads)
    return x
