"""BERT (non-autoregressive MMVID model) oracle, fp32 torch-CPU, functional over a state_dict.

TEST INFRASTRUCTURE.  Restates mmvid_pytorch/dalle_bert.py: sequence assembly and losses of
BERT.forward 869-1127, mask_predict 514-714, generate_images 434-487, get_image_tokens 716-751.
The stochastic choices of forward (mask strategy 992-1029, warp 204-238) are INJECTED
(`mask1`, `target_warp`) -- the goldens were captured after them.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import tower as T
from . import vqgan
from .axial import axial_list_table, axial_table


class Cfg:
    def __init__(self, sd, text_seq_len, num_visuals, num_targets, image_size, dim=768, use_cvae=False):
        self.dim = dim
        self.text_seq_len = text_seq_len
        self.num_visuals = num_visuals
        self.num_targets = num_targets
        self.image_size = image_size
        self.fmap = image_size // 16
        self.image_seq_len = self.fmap**2
        self.num_image_tokens = sd['vae.model.quantize.embedding.weight'].shape[0]
        # fixed language model (dalle_bert.py:307-322): no text table, ONE text token = a mapped sentence feature
        self.fixed_language_model = 'text_emb.weight' not in sd
        if self.fixed_language_model:
            assert text_seq_len == 1
        self.num_text_tokens = 1 if self.fixed_language_model else sd['text_emb.weight'].shape[0]  # already + text_seq_len
        self.visual_seq_len = num_visuals * self.image_seq_len
        self.target_seq_len = num_targets * self.image_seq_len
        self.MASK = self.num_image_tokens
        self.st1 = 1 + text_seq_len + self.visual_seq_len
        self.vid = self.st1 + 1
        self.control_seq_len = self.vid + 1
        self.total_seq_len = self.control_seq_len + self.target_seq_len
        self.use_cvae = use_cvae
        self.mask = T.build_attention_mask(self.total_seq_len, 'mask_prev', [self.st1, self.vid])


def get_image_tokens(sd, cfg, frames, which='vae'):
    """dalle_bert.py:716-751: [B,T,3,H,W] -> [B, T*n] int64."""
    b, t = frames.shape[:2]
    pref = ('cvae' if (which == 'cvae' and cfg.use_cvae) else 'vae') + '.model.'
    idx = vqgan.get_codebook_indices(sd, frames.reshape(b * t, *frames.shape[2:]), cfg.image_size, pref)
    return idx.view(b, -1)


def text_feature_mapping(sd, feat):
    """dalle_bert.py:312-322: nn.Linear, or LayerNorm-Linear-LayerNorm-Linear-LayerNorm with a bottleneck."""
    p = 'text_feature_mapping.'
    if p + 'weight' in sd:
        return F.linear(feat, sd[p + 'weight'], sd[p + 'bias'])
    h = feat
    for i in range(5):
        w, b = sd[f'{p}{i}.weight'], sd[f'{p}{i}.bias']
        h = F.layer_norm(h, (h.shape[-1], ), w, b, 1e-5) if i % 2 == 0 else F.linear(h, w, b)
    return h


def control_embedding(sd, cfg, text, visual_tok=None, with_visual=True):
    """dalle_bert.py:899-978 -> [B, 1+Ttxt+Nvis+2, dim].  with_visual=False: the `control_neg_emb` of the negvc branch, which the
    reference builds WITHOUT the visual segment even when the model has one (909-910, 927-935, 974-975)."""
    B = text.shape[0]
    sp, spp = sd['special_emb.weight'], sd['special_pos_emb.weight']
    rel = (sp[0] + spp[0]).expand(B, 1, -1)
    if cfg.fixed_language_model:  # 924-925: `text` is the feature [B, text_feature_dim]
        te = text_feature_mapping(sd, text).unsqueeze(1)
    else:
        text_range = torch.arange(cfg.text_seq_len) + (cfg.num_text_tokens - cfg.text_seq_len)
        text = torch.where(text == 0, text_range, text)  # unique pad id per position, 917-919
        te = sd['text_emb.weight'][text] + sd['text_pos_emb.weight'][:cfg.text_seq_len]
    parts = [rel, te]
    if cfg.num_visuals > 0 and with_visual:
        if visual_tok is None:
            visual_tok = torch.full((B, cfg.visual_seq_len), cfg.MASK, dtype=torch.long)
        vtab = sd['visual_emb.weight'] if 'visual_emb.weight' in sd else sd['image_emb.weight']
        vpos = axial_list_table(sd, 'visual_pos_emb', cfg.num_visuals, (cfg.fmap, cfg.fmap), cfg.dim)
        parts.append(vtab[visual_tok] + vpos)
    after = torch.stack([sp[1] + spp[1], sp[2] + spp[2]]).expand(B, 2, -1)
    parts.append(after)
    return torch.cat(parts, 1)


def target_pos(sd, cfg):
    return axial_table(sd, 'target_pos_emb', (cfg.num_targets, cfg.fmap, cfg.fmap), cfg.dim)


def head(sd, p, x):
    """nn.Sequential(LayerNorm(dim), Linear) -- dalle_bert.py:414-425."""
    h = F.layer_norm(x, (x.shape[-1], ), sd[p + '.0.weight'], sd[p + '.0.bias'], 1e-5)
    return F.linear(h, sd[p + '.1.weight'], sd[p + '.1.bias'])


def tower_fwd(sd, cfg, tokens, stable=False):
    out = T.tower(sd, tokens, cfg.mask, 'transformer.transformer.')
    if stable:  # dalle_bert.py:489-493 -> utils/utils.py:18-25 (DivideMax over the feature axis)
        out = out / out.amax(dim=-1, keepdim=True)
    return out


def forward_losses(sd, cfg, text, target_tok, mask1, warp_tok=None, visual_tok=None, rel=True, vid=True,
                   rel_no_fully_masked=True, not_fully_masked=None, stable=False, text_neg=None):
    """dalle_bert.py:1030-1127 given the injected mask / warped tokens.  Returns dict.  `text_neg` = the negvc branch
    (909-910, 927-935, 974-975, 1047-1054): the REL negative's control sequence is [REL] + text_neg + [ST1] [VID] -- never with the
    visual segment, so with visuals that pass is a SHORTER sequence (the tower slices its mask to the length, clip_model.py:218-222)."""
    B = text.shape[0]
    ctrl = control_embedding(sd, cfg, text, visual_tok)
    tpos = target_pos(sd, cfg)
    csl = cfg.control_seq_len
    if not_fully_masked is None:
        not_fully_masked = mask1.any(dim=1).float() if False else torch.ones(B)
    tm = torch.where(mask1, target_tok, cfg.MASK)
    temb = sd['image_emb.weight'][tm] + tpos
    tokens_msm = torch.cat([ctrl, temb], 1)
    out = tower_fwd(sd, cfg, tokens_msm, stable)
    logits_msm = head(sd, 'to_logits', out[:, csl:])
    loss_msm = F.cross_entropy(logits_msm[~mask1], target_tok[~mask1])
    res = dict(control_emb=ctrl, tokens_msm=tokens_msm, out_msm=out, logits_msm=logits_msm, loss_msm=loss_msm)
    nfm = not_fully_masked
    if rel:
        half = B // 2
        if text_neg is not None:
            ctrl_swap = control_embedding(sd, cfg, text_neg, with_visual=False)
        else:
            ctrl_swap = torch.cat([ctrl[half:], ctrl[:half]], 0)  # swap(): chunk(2)[::-1], 110-114
        out_neg = tower_fwd(sd, cfg, torch.cat([ctrl_swap, temb], 1), stable)
        res['tokens_rel'] = torch.cat([ctrl_swap, temb], 1)
        lp = head(sd, 'to_logits_rel', out[:, 0]).squeeze()
        ln = head(sd, 'to_logits_rel', out_neg[:, 0]).squeeze()
        if rel_no_fully_masked:
            a = F.binary_cross_entropy_with_logits(lp, torch.ones(B), reduction='none')
            b_ = F.binary_cross_entropy_with_logits(ln, torch.zeros(B), reduction='none')
            loss_rel = (a * nfm + b_ * nfm).sum() / max(1., nfm.sum())
        else:
            loss_rel = F.binary_cross_entropy_with_logits(lp, torch.ones(B)) + \
                F.binary_cross_entropy_with_logits(ln, torch.zeros(B))
        res.update(out_rel=out_neg, loss_rel=loss_rel)
    else:
        res['loss_rel'] = torch.tensor(0.0)
    if vid and cfg.num_targets > 1:
        wm = torch.where(mask1, warp_tok, cfg.MASK)
        out_neg = tower_fwd(sd, cfg, torch.cat([ctrl, sd['image_emb.weight'][wm] + tpos], 1), stable)
        lp = head(sd, 'to_logits_vid', out[:, cfg.vid])
        ln = head(sd, 'to_logits_vid', out_neg[:, cfg.vid])
        if rel_no_fully_masked:  # NB reference does not multiply by not_fully_masked here (1107-1116)
            den = max(1., nfm.sum())
            loss_vid = F.binary_cross_entropy_with_logits(lp, torch.ones(B, 1), reduction='none').sum() / den + \
                F.binary_cross_entropy_with_logits(ln, torch.zeros(B, 1), reduction='none').sum() / den
        else:
            loss_vid = F.binary_cross_entropy_with_logits(lp, torch.ones(B, 1)) + \
                F.binary_cross_entropy_with_logits(ln, torch.zeros(B, 1))
        res.update(out_vid=out_neg, loss_vid=loss_vid)
    else:
        res['loss_vid'] = torch.tensor(0.0)
    return res


def mp_schedule(cfg, mp_config, N):
    """dalle_bert.py:586-614."""
    c = mp_config
    N3 = max(1, int(N * c['N3_n']))
    N4 = max(1, int(N * c['N4_n']))
    n = list(N * np.linspace(c['N1_n'], c['N2_n'], c['T1_n'])) + list(N3 * np.ones(c['T2_n'])) + list(N4 * np.ones(c['T3_n']))
    temp = list(np.linspace(c['N1_t'], c['N2_t'], c['T1_t'])) + list(c['N3_t'] * np.ones(c['T2_t'])) + list(c['N4_t'] * np.ones(c['T3_t']))
    return list(map(int, n)), temp


def _sample_multinomial(logits, temperature):
    """dalle_bert.py:527-538 -- consumes the torch CPU generator exactly like the reference."""
    U = torch.rand_like(logits)
    g = -torch.log(-torch.log(U + 1e-20) + 1e-20)
    probs = F.softmax(logits + temperature * g, dim=2)
    b, n, c = probs.shape
    tok = torch.multinomial(probs.reshape(b * n, c), 1).view(b, n, 1)
    Y = torch.gather(probs, 2, tok)
    return Y.squeeze(2), tok.squeeze(2)


@torch.no_grad()
def mask_predict(sd, cfg, control_emb, steps, mp_config, dynamic=True):
    """dalle_bert.py:514-714 for preserve=None, long_mode='long' (the test.py path)."""
    N = cfg.target_seq_len
    csl = control_emb.shape[1]
    n, temp = mp_schedule(cfg, mp_config, N)
    Tmax = mp_config['T'] if steps <= 0 else steps
    Bm = mp_config['B']
    tpos = target_pos(sd, cfg)
    iemb = sd['image_emb.weight']
    mask_emb = iemb[cfg.MASK]
    outs = []
    for i in range(control_emb.shape[0]):
        ce = control_emb[i:i + 1]
        tok_in = torch.full((1, N), cfg.MASK, dtype=torch.long)
        out = tower_fwd(sd, cfg, torch.cat([ce, iemb[tok_in] + tpos], 1))[:, csl:]
        Y, I_tok = _sample_multinomial(head(sd, 'to_logits', out), temp[0])
        Smax, tmax, Imax = 0, 0, None
        for t in range(1, Tmax):
            embs, masks = [], []
            for j in range(Bm):
                try:
                    keep = torch.multinomial(Y.view(-1), N - n[t - 1], replacement=False)
                except RuntimeError:
                    keep = torch.multinomial(Y.view(-1), 1, replacement=False)
                m1 = torch.zeros(N).scatter_(0, keep, 1).unsqueeze(0) == 1
                masks.append(m1)
                embs.append(torch.where(m1.unsqueeze(2), iemb[I_tok], mask_emb))
            S = torch.zeros(Bm)
            YB, tokB = [], []
            for j in range(Bm):
                o = tower_fwd(sd, cfg, torch.cat([ce, embs[j] + tpos], 1))
                Yn, In = _sample_multinomial(head(sd, 'to_logits', o[:, csl:]), temp[t])
                Y = torch.where(masks[j], Y, Yn)
                I_tok = torch.where(masks[j], I_tok, In)
                s_rel = torch.sigmoid(head(sd, 'to_logits_rel', o[:, 0]))
                s_vid = torch.sigmoid(head(sd, 'to_logits_vid', o[:, cfg.vid]))
                S[j] = (s_rel * 0.5 + s_vid * 0.5).item()
                YB.append(Y)
                tokB.append(I_tok)
            jmax = S.argmax()
            Y, I_tok = YB[jmax], tokB[jmax]
            if dynamic:
                if S[jmax] > Smax:
                    tmax, Smax, Imax = t, S[jmax], I_tok
                if t - tmax >= 5:
                    break
            else:
                Imax = I_tok
        outs.append(Imax)
    return torch.cat(outs, 0)


@torch.no_grad()
def generate_images(sd, cfg, text, steps, mp_config, dynamic=True):
    """dalle_bert.py:434-487 (no visuals)."""
    ce = control_embedding(sd, cfg, text)
    seq = mask_predict(sd, cfg, ce, steps, mp_config, dynamic)
    img_seq = seq.view(-1, cfg.image_seq_len)
    images = vqgan.decode(sd, img_seq, cfg.image_size, 'vae.model.')
    return images.view(text.shape[0], cfg.num_targets, *images.shape[1:]), img_seq
