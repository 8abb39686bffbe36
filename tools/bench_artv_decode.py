#!/usr/bin/env python3
"""ART-V sampling (BASELINE config 5: 16 frames of 128x128 = 1,024 tokens, L = 1,152): seconds per video with the
reference's algorithm (full transformer over the growing prefix for every token) and with the KV-cache decoder."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvid_amd.dalle_artv import DALLE
from mmvid_amd.vae import VQGanVAE1024

dev = torch.device('cuda', 0)
torch.manual_seed(0)
vae = VQGanVAE1024(None, 128)
vae.image_size = 128
cvae = VQGanVAE1024(None, 128)
cvae.image_size = 128
m = DALLE(dim=768, vae=vae, cvae=cvae, num_text_tokens=49408, text_seq_len=64, which_transformer='openai_clip_visual',
          num_visuals=1, num_targets=16).to(dev).eval()
for B in (1, 16):
    text = torch.randint(1, 49408, (B, 64), device=dev)
    visual = torch.rand(B, 1, 3, 128, 128, device=dev)
    for cached in (True, False):
        for rep in range(2):  # the first call pays one-time costs (VQGAN plans, weight re-layout, kernel loading)
            torch.manual_seed(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            images, _, _ = m.generate_images(text, visual=visual, use_cache=cached)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f'B={B} use_cache={cached}: {dt:7.2f} s for {B} video(s) of 1024 tokens = {B * 1024 / dt:8.1f} tokens/s, images {tuple(images.shape)}')
