"""ART-V (autoregressive DALLE baseline) oracle, fp32 torch-CPU, functional.

TEST INFRASTRUCTURE.  Restates mmvid_pytorch/dalle_artv.py: DALLE.forward 418-542 (sequence
assembly, causal tower, logits over text+visual+image vocabulary, block-diagonal logits
mask 215-227/509-512, weighted CE 526-539) and generate_images 236-304 (full recompute per
step, top_k 61-67, multinomial).
"""
import torch
import torch.nn.functional as F

from . import tower as T
from . import vqgan
from .axial import axial_list_table, axial_table
from .bert import head


class Cfg:
    def __init__(self, sd, text_seq_len, num_visuals, num_targets, image_size, dim=768, loss_img_weight=7):
        self.dim, self.text_seq_len, self.num_visuals, self.num_targets = dim, text_seq_len, num_visuals, num_targets
        self.image_size = image_size
        self.fmap = image_size // 16
        self.image_seq_len = self.fmap**2
        self.target_seq_len = self.image_seq_len * num_targets
        self.visual_seq_len = self.image_seq_len * num_visuals
        self.control_seq_len = text_seq_len + self.visual_seq_len
        self.num_image_tokens = sd['image_emb.weight'].shape[0]
        self.num_text_tokens = sd['text_emb.weight'].shape[0]
        self.num_visual_tokens = sd['visual_emb.weight'].shape[0]
        self.num_control_tokens = self.num_text_tokens + self.num_visual_tokens
        self.total_tokens = self.num_control_tokens + self.num_image_tokens
        self.total_seq_len = text_seq_len + self.target_seq_len + self.visual_seq_len
        self.loss_img_weight, self.loss_vis_weight = loss_img_weight, 1.
        self.mask = T.build_attention_mask(self.total_seq_len, 'causal')
        lm = torch.block_diag(torch.ones(text_seq_len, self.num_text_tokens),
                              torch.ones(self.visual_seq_len, self.num_visual_tokens),
                              torch.ones(self.target_seq_len, self.num_image_tokens)) == 0
        self.logits_mask = lm.unsqueeze(0)


def forward(sd, cfg, text, visual_tok=None, image_tok=None, return_loss=False):
    """image_tok: [B, k] token ids (k may be 0 / None)."""
    B = text.shape[0]
    text_range = torch.arange(cfg.text_seq_len) + (cfg.num_text_tokens - cfg.text_seq_len)
    text = torch.where(text == 0, text_range, text)
    text = F.pad(text, (1, 0), value=0)  # <bos>
    tokens = sd['text_emb.weight'][text] + sd['text_pos_emb.weight'][:text.shape[1]]
    if visual_tok is None:
        visual_tok = -torch.ones(B, cfg.visual_seq_len, dtype=torch.long)
    vrange = torch.arange(cfg.visual_seq_len) + (cfg.num_visual_tokens - cfg.visual_seq_len)
    visual_tok = torch.where(visual_tok == -1, vrange, visual_tok)
    vpos = axial_list_table(sd, 'visual_pos_emb', cfg.num_visuals, (cfg.fmap, cfg.fmap), cfg.dim)
    tokens = torch.cat([tokens, sd['visual_emb.weight'][visual_tok] + vpos], 1)
    if image_tok is not None and image_tok.shape[1] > 0:
        shape = (cfg.num_targets, cfg.fmap, cfg.fmap) if cfg.num_targets > 1 else (cfg.fmap, cfg.fmap)
        ipos = axial_table(sd, 'image_pos_emb', shape, cfg.dim)
        tokens = torch.cat([tokens, sd['image_emb.weight'][image_tok] + ipos[:image_tok.shape[1]]], 1)
    if tokens.shape[1] > cfg.total_seq_len:
        tokens = tokens[:, :-1]
    seq_len = tokens.shape[1]
    out = T.tower(sd, tokens, cfg.mask, 'transformer.transformer.')
    logits = head(sd, 'to_logits', out)
    logits = logits.masked_fill(cfg.logits_mask[:, :seq_len], -torch.finfo(logits.dtype).max)
    if not return_loss:
        return logits
    labels = torch.cat([text[:, 1:], visual_tok + cfg.num_text_tokens, image_tok + cfg.num_control_tokens], 1)
    lg = logits.permute(0, 2, 1)
    tl, cl = cfg.text_seq_len, cfg.control_seq_len
    loss_text = F.cross_entropy(lg[:, :, :tl], labels[:, :tl])
    loss_vis = F.cross_entropy(lg[:, :, tl:cl], labels[:, tl:cl])
    loss_img = F.cross_entropy(lg[:, :, cl:], labels[:, cl:])
    return (loss_text + cfg.loss_vis_weight * loss_vis + cfg.loss_img_weight * loss_img) / (cfg.loss_img_weight + cfg.loss_vis_weight + 1)


def top_k(logits, thres=0.5):
    k = max(int((1 - thres) * logits.shape[-1]), 1)
    val, ind = torch.topk(logits, k)
    return torch.full_like(logits, float('-inf')).scatter_(1, ind, val)


@torch.no_grad()
def generate_images(sd, cfg, text, visual_tok=None, filter_thres=0.5, temperature=1.):
    """dalle_artv.py:236-304: one full forward per generated token."""
    out = text[:, :cfg.text_seq_len]
    for cur in range(out.shape[1], cfg.text_seq_len + cfg.target_seq_len):
        image = out[:, cfg.text_seq_len:]
        logits = forward(sd, cfg, out[:, :cfg.text_seq_len], visual_tok, image)[:, -1, :]
        probs = F.softmax(top_k(logits, filter_thres) / temperature, dim=-1)
        sample = torch.multinomial(probs, 1) - cfg.num_control_tokens
        out = torch.cat([out, sample], -1)
    img_seq = out[:, -cfg.target_seq_len:].reshape(-1, cfg.image_seq_len)
    images = vqgan.decode(sd, img_seq, cfg.image_size, 'vae.model.')
    return images.view(text.shape[0], cfg.num_targets, *images.shape[1:]), img_seq
