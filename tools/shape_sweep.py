"""Does every public entry point run at every batch size / frame count a user may pass?  A sweep over shapes (2-layer towers,
full-size VQGAN) that only checks for exceptions and non-finite outputs -- parity lives in tests/.  python tools/shape_sweep.py"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmvid_amd.engine import FlatTrainer, backward_order  # noqa: E402

dev = torch.device('cuda', 0)
fails = []


def run(name, fn):
    try:
        out = fn()
        torch.cuda.synchronize()
        ts = [t for t in (out if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(t) and t.is_floating_point()]
        bad = [tuple(t.shape) for t in ts if not torch.isfinite(t).all()]
        print(('ok   ' if not bad else 'NONFINITE ') + name, bad if bad else '', flush=True)
        if bad:
            fails.append(name)
    except Exception as e:  # noqa
        print('FAIL ' + name + ': ' + (str(e).splitlines() or [''])[0][:200], flush=True)
        if os.environ.get('SWEEP_TRACE'):
            traceback.print_exc()
        fails.append(name)


gen = torch.Generator().manual_seed(0)
for cfg in (2, 4):
    torch.manual_seed(1)
    m = bench.build_model(cfg, dev, 2)
    tr = FlatTrainer(m, lr=1e-4, max_grad_norm=1.0, order=backward_order)
    for B in (2, 4, 6, 8, 10):
        b = bench.synth_batch(B, 8, dev, gen, visuals=1 if cfg == 4 else 0)
        kw = dict(visual=b['visual']) if cfg == 4 else {}

        def train():
            m.train()
            tr.zero_grad()
            lm, lr, lv = m(b['text'], target=b['frames'], return_loss=True, rel=True, vid=True, rel_no_fully_masked=True, **kw)
            (lm + lr + lv).backward()
            tr.step()
            return lm, lr, lv
        run(f'BERT config {cfg} train B={B}', train)
    for B in (1, 3):
        b = bench.synth_batch(B, 8, dev, gen, visuals=1 if cfg == 4 else 0)
        kw = dict(visual=b['visual']) if cfg == 4 else {}
        m.train()
        run(f'BERT config {cfg} train B={B} (odd: MSM only)', lambda: m(b['text'], target=b['frames'], return_loss=True, **kw))
    m.eval()
    for B, cand, dyn in ((1, 1, True), (3, 1, False), (5, 2, True), (16, 1, False), (2, 3, True)):
        b = bench.synth_batch(B, 8, dev, gen, visuals=1 if cfg == 4 else 0)
        kw = dict(visual=b['visual']) if cfg == 4 else {}
        run(f'BERT config {cfg} generate_images b={B} candidates={cand} dynamic={dyn}',
            lambda: m.generate_images(b['text'], mask_predict_steps=0, mp_config=dict(bench.MP_CONFIG, B=cand, T=6), dynamic=dyn, **kw)[0])
    vae = m.vae
    for strict in (False, True, 'split'):
        vae.strict = strict
        for N in (1, 2, 7, 54) if strict is not True else (1, 3):
            img = torch.rand(N, 3, 128, 128, device=dev)
            run(f'VQGAN encode+decode N={N} strict={strict}', lambda: vae.decode(vae.get_codebook_indices(img)))
        vae.strict = False
    del m, tr
torch.manual_seed(2)
a = bench.build_model(5, dev, 2).eval()
for B in (1, 3, 5, 9):
    text = torch.randint(1, 49408, (B, 64), device=dev)
    vis = torch.randint(0, 1024, (B, 64), device=dev)
    for cache in (True, False) if B <= 3 else (True, ):
        run(f'ART-V generate_images b={B} use_cache={cache}', lambda: a.generate_images(text, visual=vis, use_cache=cache)[0])
    tt = torch.randint(0, 1024, (B, 1024), device=dev)
    a.train()
    run(f'ART-V train forward b={B}', lambda: a(text, visual=vis, target=tt, return_loss=True)[0])
    a.eval()
print('failures:', fails)
sys.exit(1 if fails else 0)
