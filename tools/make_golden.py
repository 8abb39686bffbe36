#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference) on CPU.

Build-container only: the reference is imported (never copied) through tools/ref_stubs.py;
weights and inputs are the deterministic synthetic ones of oracle/synth.py, so a fixture
holds seeds, the reference's state_dict manifest (key -> shape: pins the checkpoint
layout), and the reference's outputs.  Run:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [case ...]

Cases: vq vqgan_tiny vqgan_full vqgan_full16 vqgan_full16_refinit tower tower12 bert_tiny bert_tiny_visual bert_negvc bert_negvc_visual bert_flm bert_flm_bottleneck artv_tiny mask_predict
       frontend mask_predict_race
"""
import json
import os
import random
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

import torch  # noqa: E402

import ref_stubs  # noqa: E402
from oracle.synth import synth_input, synth_state_dict, synth_tensor, synth_tokens  # noqa: E402

ref_stubs.install()  # chdir -> /root/reference, stub modules
torch.set_num_threads(8)

OUT = os.path.join(REPO, 'tests', 'golden')
MP_CONFIG = dict(T1_n=10, T2_n=10, T3_n=30, N1_n=0.9, N2_n=0.1, N3_n=0.125, N4_n=0.0625,
                 T1_t=10, T2_t=5, T3_t=35, N1_t=0., N2_t=0., N3_t=0., N4_t=0., T=20, B=1)


def manifest_of(module):
    return [(k, list(v.shape)) for k, v in module.state_dict().items()]


def load_synth(module, seed):
    man = manifest_of(module)
    module.load_state_dict(synth_state_dict(man, seed))
    return man


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        if isinstance(v, (dict, list)) and not isinstance(v, np.ndarray):
            v = np.frombuffer(json.dumps(v).encode(), dtype=np.uint8)
        out[k] = v
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}  {os.path.getsize(path) / 1024:.1f} KiB')


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


# --------------------------------------------------------------------------- builders
def build_vae(tiny, seed):
    from mmvid_pytorch.vae import VQGanVAE1024
    if tiny:
        ref_stubs.VQGAN_OVERRIDES.update(n_embed=256, ch=32)
        image_size = 64
    else:
        ref_stubs.VQGAN_OVERRIDES.clear()
        image_size = 128
    vae = VQGanVAE1024(None, image_size)
    vae.image_size = image_size  # the driver does this, train.py:182-185
    if tiny:
        vae.num_tokens = 256
    man = load_synth(vae, seed)
    vae.eval()
    return vae, man


def clip_state(layers):
    from mmvid_pytorch.transformers import clip_model
    clip = clip_model.CLIP(512, 224, layers, 768, 32, 77, 49408, 512, 8, layers)
    return clip.state_dict()


# ------------------------------------------------------------------------------ cases
def case_vq():
    from taming.modules.vqvae.quantize import VectorQuantizer2
    res = {}
    for tag, n_e in (('sep', 1024), ('stress', 1024), ('small', 256)):
        q = VectorQuantizer2(n_e, 256, beta=0.25)
        if tag == 'stress':  # reference's own init range U(-1/n, 1/n): near-tie stress
            cb = (synth_input('cb_' + tag, (n_e, 256), 7, 'uniform') * 2 - 1) / n_e
        else:
            cb = synth_tensor('quantize.embedding.weight', (n_e, 256), 7)
        q.embedding.weight.data.copy_(cb)
        z = synth_input('z_' + tag, (512, 256), 7)
        z4 = z.view(8, 8, 8, 256).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            zq, _, (_, _, idx) = q(z4)
            d = torch.sum(z**2, dim=1, keepdim=True) + torch.sum(cb**2, dim=1) - 2 * (z @ cb.t())
        top2 = torch.topk(d, 2, dim=1, largest=False)
        res[tag + '_idx'] = idx.view(-1)
        res[tag + '_top2_d'] = top2.values
        res[tag + '_top2_i'] = top2.indices
        res[tag + '_zq_sum'] = zq.double().sum().view(1)
    save('vq', meta=dict(seed=7, rows=512, e_dim=256), **res)


def _vqgan(name, tiny, n):
    vae, man = build_vae(tiny, 11)
    s = vae.image_size
    img = synth_input('img', (n, 3, s, s), 11, 'uniform')
    with torch.no_grad():
        h = vae.model.quant_conv(vae.model.encoder(2 * img - 1))
        idx = vae.get_codebook_indices(img)
        dec = vae.decode(idx)
        # distance margins of the reference expression, for near-tie analysis
        cb = vae.model.quantize.embedding.weight
        zf = h.permute(0, 2, 3, 1).reshape(-1, cb.shape[1])
        d = torch.sum(zf**2, 1, keepdim=True) + torch.sum(cb**2, 1) - 2 * zf @ cb.t()
        top2 = torch.topk(d, 2, dim=1, largest=False)
    save(name, meta=dict(seed=11, n=n, image_size=s, tiny=tiny), manifest=man, z_e=h, indices=idx,
         top2_d=top2.values, decoded=dec)


def case_vqgan_tiny():
    _vqgan('vqgan_tiny', True, 2)


def case_vqgan_full():
    _vqgan('vqgan_full', False, 1)


def _vqgan_wide(name, seed, n, codebook):
    """Round 6: the index contract on more than one frame.  n full-size frames through the reference's encoder + quantiser on the
    synthetic weights of `seed`; `codebook` = 'synthetic' keeps oracle.synth's 0.5 N(0,1) rows (well separated), 'reference_init'
    overwrites them with the reference's own initialisation U(-1/n_e, 1/n_e) (quantize.py:254: near-uniform, the near-tie stress
    case of SURVEY 8c; the rows are a function of the seed, see `wide_codebook` in tests/conftest.py).  Kept per token: index, the
    two smallest distances of the reference expression and their codes; z_e for the first 4 frames only (the file stays < 1 MB)."""
    vae, man = build_vae(False, seed)
    cb = vae.model.quantize.embedding.weight
    if codebook == 'reference_init':
        cb.data.copy_((synth_input('codebook_reference_init', tuple(cb.shape), seed, 'uniform') * 2 - 1) / cb.shape[0])
    img = synth_input('img_wide', (n, 3, 128, 128), seed, 'uniform')
    with torch.no_grad():
        h = torch.cat([vae.model.quant_conv(vae.model.encoder(2 * img[i:i + 4] - 1)) for i in range(0, n, 4)])
        idx = torch.cat([vae.get_codebook_indices(img[i:i + 4]) for i in range(0, n, 4)])
        zf = h.permute(0, 2, 3, 1).reshape(-1, cb.shape[1])
        d = torch.sum(zf**2, 1, keepdim=True) + torch.sum(cb**2, 1) - 2 * zf @ cb.t()
        top2 = torch.topk(d, 2, dim=1, largest=False)
        dec = vae.decode(idx[:1])
    assert torch.equal(top2.indices[:, 0].view_as(idx), idx) or codebook == 'reference_init'
    gap = (top2.values[:, 1] - top2.values[:, 0])
    print(f'{name}: {idx.numel()} tokens, {idx.unique().numel()} distinct codes; top-2 gap min {gap.min().item():.3e} '
          f'median {gap.median().item():.3e}; |z| rms {zf.pow(2).mean().sqrt().item():.3f}')
    save(name, meta=dict(seed=seed, n=n, image_size=128, tiny=False, codebook=codebook, z_frames=4), manifest=man,
         z_e=h[:4], indices=idx, top2_d=top2.values, top2_i=top2.indices.to(torch.int16), decoded=dec)


def case_vqgan_full16():
    _vqgan_wide('vqgan_full16', 11, 16, 'synthetic')


def case_vqgan_full16_refinit():
    _vqgan_wide('vqgan_full16_refinit', 23, 16, 'reference_init')


def case_tower():
    from mmvid_pytorch.transformers.clip_model import OpenAICLIPTransformer
    ref_stubs.CLIP_STATE['sd'] = clip_state(2)
    res = {}
    man = None
    for tag, L, mtype, kw in (('L51', 51, 'mask_prev', {'index': [17, 18]}),
                              ('L579', 579, 'mask_prev', {'index': [65, 66]}),
                              ('causal40', 40, 'causal', {})):
        tw = OpenAICLIPTransformer(L, 'openai_clip_visual', model_path='x', causal=True,
                                   mask_type=mtype, mask_kwargs=kw)
        man = load_synth(tw, 13)
        x = synth_input('x_' + tag, (2, L, 768), 13).requires_grad_(True)
        g = synth_input('g_' + tag, (2, L, 768), 13)
        y = tw(x)
        y.backward(g)
        blk = tw.transformer.resblocks
        if L <= 64:
            res[tag + '_y'] = y
            res[tag + '_dx'] = x.grad
        else:
            res[tag + '_y_s'] = y[:, ::37, ::13]
            res[tag + '_dx_s'] = x.grad[:, ::37, ::13]
        res[tag + '_y_norm'] = y.double().norm().view(1)
        res[tag + '_dx_norm'] = x.grad.double().norm().view(1)
        for nm, p in (('inw', blk[0].attn.in_proj_weight), ('outw', blk[1].attn.out_proj.weight),
                      ('fcw', blk[0].mlp.c_fc.weight), ('pjw', blk[1].mlp.c_proj.weight)):
            res[f'{tag}_d{nm}_s'] = p.grad[::61, ::29]
            res[f'{tag}_d{nm}_norm'] = p.grad.double().norm().view(1)
        for nm, p in (('inb', blk[0].attn.in_proj_bias), ('ln1w', blk[0].ln_1.weight),
                      ('ln2b', blk[1].ln_2.bias), ('fcb', blk[1].mlp.c_fc.bias)):
            res[f'{tag}_d{nm}'] = p.grad
    save('tower', meta=dict(seed=13, layers=2, width=768, heads=12), manifest=man, **res)


def case_tower12():
    """The full-depth tower at the training shape (12 layers, L = 579, mask_prev rows 65 / 66): slices and norms of the output,
    the input gradient and parameter gradients of the first, a middle and the last layer."""
    from mmvid_pytorch.transformers.clip_model import OpenAICLIPTransformer
    ref_stubs.CLIP_STATE['sd'] = clip_state(12)
    L = 579
    tw = OpenAICLIPTransformer(L, 'openai_clip_visual', model_path='x', causal=True, mask_type='mask_prev',
                               mask_kwargs={'index': [65, 66]})
    man = load_synth(tw, 23)
    assert len(tw.transformer.resblocks) == 12
    x = synth_input('x_t12', (2, L, 768), 23).requires_grad_(True)
    g = synth_input('g_t12', (2, L, 768), 23)
    y = tw(x)
    y.backward(g)
    res = {'y_s': y[:, ::37, ::13], 'dx_s': x.grad[:, ::37, ::13], 'y_norm': y.double().norm().view(1),
           'dx_norm': x.grad.double().norm().view(1), 'y_rows': y[:, [0, 64, 65, 66, 67, 578]][..., ::7],
           'dx_rows': x.grad[:, [0, 64, 65, 66, 67, 578]][..., ::7]}
    blk = tw.transformer.resblocks
    for li in (0, 5, 11):
        for nm, p in (('inw', blk[li].attn.in_proj_weight), ('outw', blk[li].attn.out_proj.weight),
                      ('fcw', blk[li].mlp.c_fc.weight), ('pjw', blk[li].mlp.c_proj.weight)):
            res[f'l{li}_d{nm}_s'] = p.grad[::61, ::29]
            res[f'l{li}_d{nm}_norm'] = p.grad.double().norm().view(1)
        for nm, p in (('inb', blk[li].attn.in_proj_bias), ('ln1w', blk[li].ln_1.weight), ('ln2b', blk[li].ln_2.bias),
                      ('fcb', blk[li].mlp.c_fc.bias), ('pjb', blk[li].mlp.c_proj.bias)):
            res[f'l{li}_d{nm}'] = p.grad
    save('tower12', meta=dict(seed=23, layers=12, width=768, heads=12, L=L), manifest=man, **res)


def _build_bert(num_visuals, use_cvae, seed, text_seq_len=16, num_targets=2, **extra):
    from mmvid_pytorch.dalle_bert import BERT
    ref_stubs.CLIP_STATE['sd'] = clip_state(2)
    vae, _ = build_vae(True, 11)
    cvae = build_vae(True, 12)[0] if use_cvae else None
    m = BERT(dim=768, vae=vae, cvae=cvae, num_text_tokens=49408, text_seq_len=text_seq_len,
             which_transformer='openai_clip_visual', num_visuals=num_visuals, num_targets=num_targets,
             openai_clip_path='x', **extra)
    man = manifest_of(m)
    sd = synth_state_dict(man, seed)
    # keep the per-VAE seeds used above so vae / cvae differ
    for k in list(sd):
        if k.startswith('vae.'):
            sd[k] = synth_tensor(k[len('vae.'):], sd[k].shape, 11)
        elif k.startswith('cvae.'):
            sd[k] = synth_tensor(k[len('cvae.'):], sd[k].shape, 12)
    m.load_state_dict(sd)
    return m, man


def _bert_case(name, num_visuals, use_cvae):
    import mmvid_pytorch.dalle_bert as db
    m, man = _build_bert(num_visuals, use_cvae, 17)
    B, T, S, TL = 2, 2, 64, 16
    text = synth_tokens('text', (B, TL), 49408, 17, low=1)
    text[0, 11:] = 0
    text[1, 5:] = 0
    frames = synth_input('frames', (B, T, 3, S, S), 17, 'uniform')
    visual = synth_input('visual', (B, num_visuals, 3, S, S), 17, 'uniform') if num_visuals else None

    cap = {'emb_in': [], 'tf_in': [], 'tf_out': [], 'warp': []}
    
    def emb_hook(mod, i, o):
        cap['emb_in'].append(i[0].clone())

    h1 = m.image_emb.register_forward_hook(emb_hook)

    def tf_hook(mod, i, o):
        cap['tf_in'].append(i[0].detach().clone())
        cap['tf_out'].append(o.detach().clone())

    h2 = m.transformer.register_forward_hook(tf_hook)
    warp_orig = db.warp

    def warp_cap(x, p):
        y = warp_orig(x, p)
        cap['warp'].append(y.clone())
        return y

    db.warp = warp_cap
    m.train()
    with torch.no_grad():
        ctrl = m(text, visual=visual, return_loss=False)
    seed_all(123)
    loss_msm, loss_rel, loss_vid = m(text, visual=visual, target=frames, return_loss=True, rel=True, vid=True,
                                     msm_strategy_prob=np.array([0.7, 0.1, 0.1, 0.1]),
                                     msm_bernoulli_prob=[0.2, 0.5], rel_no_fully_masked=True,
                                     vid_strategy_prob=np.array([0.25, 0.25, 0.25, 0.25]))
    loss = 7 * loss_msm + 0.5 * loss_rel + 0.5 * loss_vid
    loss.backward()
    db.warp = warp_orig
    h1.remove(), h2.remove()
    with torch.no_grad():
        target_tok = m.get_image_tokens(frames)
        warp_tok = m.get_image_tokens(cap['warp'][0])
    target_masked, warp_masked = cap['emb_in'][0], cap['emb_in'][1]
    mask1 = target_masked != m.image_token_lut['[MASK]']
    assert torch.equal(torch.where(mask1, target_tok, 1024 * 0 + m.image_token_lut['[MASK]']), target_masked)
    csl = ctrl.shape[1]
    with torch.no_grad():
        logits_msm = m.to_logits(cap['tf_out'][0][:, csl:])
    g = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    res = dict(text=text, frames=frames, control_emb=ctrl, target_tok=target_tok, warp_tok=warp_tok, mask1=mask1,
               warped_frames=cap['warp'][0], tokens_msm=cap['tf_in'][0], out_msm=cap['tf_out'][0],
               out_rel_s=cap['tf_out'][1][:, ::3, ::7], out_vid_s=cap['tf_out'][2][:, ::3, ::7],
               logits_msm=logits_msm, losses=torch.stack([loss_msm, loss_rel, loss_vid]).detach(),
               g_image_emb=g['image_emb.weight'][::3, ::5], g_to_logits_w=g['to_logits.1.weight'][::4, ::6],
               g_special_emb=g['special_emb.weight'], g_text_pos=g['text_pos_emb.weight'][:, ::5],
               g_tpos0=g['target_pos_emb.weights_0'].reshape(-1, 768), g_tpos2=g['target_pos_emb.weights_2'].reshape(-1, 768),
               g_ln1w=g['transformer.transformer.resblocks.0.ln_1.weight'],
               g_fcb=g['transformer.transformer.resblocks.1.mlp.c_fc.bias'],
               g_relw=g['to_logits_rel.1.weight'], g_vidw=g['to_logits_vid.1.weight'],
               g_text_emb_norm=g['text_emb.weight'].double().norm().view(1),
               g_text_emb_rows=g['text_emb.weight'][text.view(-1).unique()][:, ::11],
               g_text_emb_row_ids=text.view(-1).unique(),
               g_total_norm=torch.sqrt(sum((v.double()**2).sum() for v in g.values())).view(1))
    if visual is not None:
        res['visual'] = visual
        with torch.no_grad():
            res['visual_tok'] = m.get_image_tokens(visual, which_vae='cvae')
        res['g_visual_emb'] = g['visual_emb.weight'][::3, ::5]
        res['g_vpos0'] = g['visual_pos_emb.module_list.0.weights_0'].reshape(-1, 768)
    save(name, meta=dict(seed=17, vae_seed=11, cvae_seed=12, B=B, T=T, image_size=S, text_seq_len=TL,
                         num_visuals=num_visuals, layers=2, py_seed=123), manifest=man, **res)


def case_bert_negvc():
    """BERT.forward with negvc=True, no visuals (dalle_bert.py:909-910, 927-935, 974-975, 1047-1065): the REL negative is the
    control sequence of ANOTHER caption (`text_neg`) instead of the batch's swapped halves."""
    import mmvid_pytorch.dalle_bert as db
    m, man = _build_bert(0, False, 17)
    B, T, S, TL = 2, 2, 64, 16
    text = synth_tokens('text', (B, TL), 49408, 17, low=1)
    text[0, 11:] = 0
    text[1, 5:] = 0
    text_neg = synth_tokens('text_neg', (B, TL), 49408, 17, low=1)
    text_neg[0, 7:] = 0
    text_neg[1, 13:] = 0
    frames = synth_input('frames', (B, T, 3, S, S), 17, 'uniform')
    cap = {'emb_in': [], 'tf_in': [], 'tf_out': [], 'warp': []}
    h1 = m.image_emb.register_forward_hook(lambda mod, i, o: cap['emb_in'].append(i[0].clone()))

    def tf_hook(mod, i, o):
        cap['tf_in'].append(i[0].detach().clone())
        cap['tf_out'].append(o.detach().clone())

    h2 = m.transformer.register_forward_hook(tf_hook)
    warp_orig = db.warp

    def warp_cap(x, p):
        y = warp_orig(x, p)
        cap['warp'].append(y.clone())
        return y

    db.warp = warp_cap
    m.train()
    seed_all(123)
    loss_msm, loss_rel, loss_vid = m(text, target=frames, return_loss=True, rel=True, vid=True, negvc=True, text_neg=text_neg.clone(),
                                     msm_strategy_prob=np.array([0.7, 0.1, 0.1, 0.1]), msm_bernoulli_prob=[0.2, 0.5],
                                     rel_no_fully_masked=True, vid_strategy_prob=np.array([0.25, 0.25, 0.25, 0.25]))
    (7 * loss_msm + 0.5 * loss_rel + 0.5 * loss_vid).backward()
    db.warp = warp_orig
    h1.remove(), h2.remove()
    with torch.no_grad():
        target_tok = m.get_image_tokens(frames)
        warp_tok = m.get_image_tokens(cap['warp'][0])
    mask1 = cap['emb_in'][0] != m.image_token_lut['[MASK]']
    g = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    ids = torch.cat((text.view(-1), text_neg.view(-1))).unique()
    ids = ids[ids != 0]
    save('bert_negvc', meta=dict(seed=17, vae_seed=11, B=B, T=T, image_size=S, text_seq_len=TL, num_visuals=0, layers=2, py_seed=123),
         manifest=man, text=text, text_neg=text_neg, frames=frames, target_tok=target_tok, warp_tok=warp_tok, mask1=mask1,
         warped_frames=cap['warp'][0], tokens_rel_ctrl=cap['tf_in'][1][:, :19], out_msm_s=cap['tf_out'][0][:, ::3, ::7],
         out_rel_s=cap['tf_out'][1][:, ::3, ::7], losses=torch.stack([loss_msm, loss_rel, loss_vid]).detach(),
         g_special_emb=g['special_emb.weight'], g_text_pos=g['text_pos_emb.weight'][:, ::5],
         g_relw=g['to_logits_rel.1.weight'], g_text_emb_rows=g['text_emb.weight'][ids][:, ::11], g_text_emb_row_ids=ids,
         g_total_norm=torch.sqrt(sum((v.double()**2).sum() for v in g.values())).view(1))


def case_bert_negvc_visual():
    """Round 6: negvc=True TOGETHER with a visual control (dalle_bert.py:908-909, 927-935, 974-975, 1047-1054; reachable from
    train.py:312-314 with --negvc --visual).  The reference builds control_neg = [REL] + text_neg + ([ST1], [VID]) WITHOUT the visual
    segment and `visual_neg` is accepted and ignored: the REL-negative pass is a SHORTER sequence (the attention mask is sliced to its
    length, clip_model.py:218-222, so the restricted rows then fall on target tokens).  Captured: that pass's input / output slices."""
    import mmvid_pytorch.dalle_bert as db
    m, man = _build_bert(1, True, 17)
    B, T, S, TL = 2, 2, 64, 16
    text = synth_tokens('text', (B, TL), 49408, 17, low=1)
    text[0, 11:] = 0
    text[1, 5:] = 0
    text_neg = synth_tokens('text_neg', (B, TL), 49408, 17, low=1)
    text_neg[0, 7:] = 0
    text_neg[1, 13:] = 0
    frames = synth_input('frames', (B, T, 3, S, S), 17, 'uniform')
    visual = synth_input('visual', (B, 1, 3, S, S), 17, 'uniform')
    visual_neg = synth_input('visual_neg', (B, 1, 3, S, S), 17, 'uniform')  # (ignored by the reference)
    cap = {'emb_in': [], 'tf_in': [], 'tf_out': [], 'warp': []}
    h1 = m.image_emb.register_forward_hook(lambda mod, i, o: cap['emb_in'].append(i[0].clone()))

    def tf_hook(mod, i, o):
        cap['tf_in'].append(i[0].detach().clone())
        cap['tf_out'].append(o.detach().clone())

    h2 = m.transformer.register_forward_hook(tf_hook)
    warp_orig = db.warp

    def warp_cap(x, p):
        y = warp_orig(x, p)
        cap['warp'].append(y.clone())
        return y

    db.warp = warp_cap
    m.train()
    seed_all(123)
    loss_msm, loss_rel, loss_vid = m(text, visual=visual, target=frames, return_loss=True, rel=True, vid=True, negvc=True,
                                     text_neg=text_neg.clone(), visual_neg=visual_neg,
                                     msm_strategy_prob=np.array([0.7, 0.1, 0.1, 0.1]), msm_bernoulli_prob=[0.2, 0.5],
                                     rel_no_fully_masked=True, vid_strategy_prob=np.array([0.25, 0.25, 0.25, 0.25]))
    (7 * loss_msm + 0.5 * loss_rel + 0.5 * loss_vid).backward()
    db.warp = warp_orig
    h1.remove(), h2.remove()
    with torch.no_grad():
        target_tok = m.get_image_tokens(frames)
        warp_tok = m.get_image_tokens(cap['warp'][0])
        visual_tok = m.get_image_tokens(visual, which_vae='cvae')
    mask1 = cap['emb_in'][0] != m.image_token_lut['[MASK]']
    g = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    ids = torch.cat((text.view(-1), text_neg.view(-1))).unique()
    ids = ids[ids != 0]
    lens = [int(t.shape[1]) for t in cap['tf_in']]
    print('tower passes (sequence lengths):', lens)
    save('bert_negvc_visual', meta=dict(seed=17, vae_seed=11, cvae_seed=12, B=B, T=T, image_size=S, text_seq_len=TL, num_visuals=1, layers=2,
                                        py_seed=123, pass_lengths=lens),
         manifest=man, text=text, text_neg=text_neg, frames=frames, visual=visual, visual_neg=visual_neg, target_tok=target_tok,
         warp_tok=warp_tok, visual_tok=visual_tok, mask1=mask1, warped_frames=cap['warp'][0],
         tokens_rel=cap['tf_in'][1][:, :, ::5], out_msm_s=cap['tf_out'][0][:, ::3, ::7], out_rel=cap['tf_out'][1][:, :, ::7],
         out_vid_s=cap['tf_out'][2][:, ::3, ::7], losses=torch.stack([loss_msm, loss_rel, loss_vid]).detach(),
         g_special_emb=g['special_emb.weight'], g_text_pos=g['text_pos_emb.weight'][:, ::5], g_relw=g['to_logits_rel.1.weight'],
         g_visual_emb=g['visual_emb.weight'][::3, ::5], g_text_emb_rows=g['text_emb.weight'][ids][:, ::11], g_text_emb_row_ids=ids,
         g_fcb=g['transformer.transformer.resblocks.1.mlp.c_fc.bias'],
         g_total_norm=torch.sqrt(sum((v.double()**2).sum() for v in g.values())).view(1))


def _bert_flm_case(name, bottleneck):
    """BERT with a fixed language model (dalle_bert.py:307-322, 924-925): the text is one sentence feature per sample (what
    utils_train.py:194-215 takes from RoBERTa-large, 1024 wide); the model maps it to one token."""
    import mmvid_pytorch.dalle_bert as db
    FD = 1024
    m, man = _build_bert(0, False, 23, fixed_language_model='roberta-large', text_feature_dim=FD, text_emb_bottleneck=bottleneck)
    B, T, S = 2, 2, 64
    feat = synth_input('text_feat', (B, FD), 23, 'normal')
    frames = synth_input('frames', (B, T, 3, S, S), 23, 'uniform')
    cap = {'emb_in': [], 'tf_out': [], 'warp': []}
    h1 = m.image_emb.register_forward_hook(lambda mod, i, o: cap['emb_in'].append(i[0].clone()))
    h2 = m.transformer.register_forward_hook(lambda mod, i, o: cap['tf_out'].append(o.detach().clone()))
    warp_orig = db.warp

    def warp_cap(x, p):
        y = warp_orig(x, p)
        cap['warp'].append(y.clone())
        return y

    db.warp = warp_cap
    m.train()
    with torch.no_grad():
        ctrl = m(feat, return_loss=False)
    seed_all(321)
    losses = m(feat, target=frames, return_loss=True, rel=True, vid=True, msm_strategy_prob=np.array([0.7, 0.1, 0.1, 0.1]),
               msm_bernoulli_prob=[0.2, 0.5], rel_no_fully_masked=True, vid_strategy_prob=np.array([0.25, 0.25, 0.25, 0.25]))
    (7 * losses[0] + 0.5 * losses[1] + 0.5 * losses[2]).backward()
    db.warp = warp_orig
    h1.remove(), h2.remove()
    with torch.no_grad():
        target_tok = m.get_image_tokens(frames)
        warp_tok = m.get_image_tokens(cap['warp'][0])
    mask1 = cap['emb_in'][0] != m.image_token_lut['[MASK]']
    g = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    res = dict(text_feat=feat, frames=frames, control_emb=ctrl, target_tok=target_tok, warp_tok=warp_tok, mask1=mask1,
               out_msm_s=cap['tf_out'][0][:, ::3, ::7], out_rel_s=cap['tf_out'][1][:, ::3, ::7], out_vid_s=cap['tf_out'][2][:, ::3, ::7],
               losses=torch.stack(list(losses)).detach(), g_image_emb=g['image_emb.weight'][::3, ::5],
               g_special_emb=g['special_emb.weight'],
               g_total_norm=torch.sqrt(sum((v.double()**2).sum() for v in g.values())).view(1))
    for k, v in g.items():
        if k.startswith('text_feature_mapping.'):
            res['g_' + k] = v if v.dim() == 1 else v[::4, ::8]
            res['gn_' + k] = v.double().norm().view(1)
    save(name, meta=dict(seed=23, vae_seed=11, B=B, T=T, image_size=S, text_feature_dim=FD, bottleneck=bottleneck, layers=2,
                         py_seed=321), manifest=man, **res)


def case_bert_flm():
    _bert_flm_case('bert_flm', None)


def case_bert_flm_bottleneck():
    _bert_flm_case('bert_flm_bottleneck', '256')


def case_bert_tiny():
    _bert_case('bert_tiny', 0, False)


def case_bert_tiny_visual():
    _bert_case('bert_tiny_visual', 1, True)


def case_artv_tiny():
    from mmvid_pytorch.dalle_artv import DALLE
    ref_stubs.CLIP_STATE['sd'] = clip_state(2)
    vae, _ = build_vae(True, 11)
    m = DALLE(dim=768, vae=vae, cvae=None, num_text_tokens=49408, text_seq_len=16,
              which_transformer='openai_clip_visual', num_visuals=1, num_targets=2, openai_clip_path='x')
    man = manifest_of(m)
    sd = synth_state_dict(man, 19)
    for k in list(sd):
        if k.startswith('vae.'):
            sd[k] = synth_tensor(k[len('vae.'):], sd[k].shape, 11)
    m.load_state_dict(sd)
    B = 2
    text = synth_tokens('text', (B, 16), 49408, 19, low=1)
    text[0, 9:] = 0
    frames = synth_input('frames', (B, 2, 3, 64, 64), 19, 'uniform')
    visual = synth_input('visual', (B, 1, 3, 64, 64), 19, 'uniform')
    res = dict(text=text, frames=frames, visual=visual)
    m.train()
    loss, _, _ = m(text, visual=visual, target=frames, return_loss=True)
    loss.backward()
    res['loss'] = loss.detach().view(1)
    res['g_to_logits_w_s'] = m.to_logits[1].weight.grad[::997, ::13]
    res['g_image_emb_s'] = m.image_emb.weight.grad[::3, ::5]
    res['g_total_norm'] = torch.sqrt(sum((p.grad.double()**2).sum() for p in m.parameters() if p.grad is not None)).view(1)
    loss_nv, _, _ = m(text, visual=None, target=frames, return_loss=True)
    res['loss_novisual'] = loss_nv.detach().view(1)
    with torch.no_grad():
        tok = m.get_image_tokens(frames)
        res['target_tok'] = tok
        for k in (0, 5, 31):
            lg = m(text, visual=visual, target=tok[:, :k])
            last = lg[:, -1, :]
            res[f'logits_k{k}_s'] = last[:, ::97]
            res[f'logits_k{k}_lse'] = torch.logsumexp(last.double(), -1)
            res[f'logits_k{k}_argmax'] = last.argmax(-1)
            res[f'logits_k{k}_img'] = last[:, m.num_control_tokens:]
    # sampling trajectory under the CPU generator (oracle must consume RNG identically)
    m.eval()
    seed_all(5)
    images, _, _ = m.generate_images(text[:1], visual=visual[:1])
    res['gen_images_sum'] = images.double().sum().view(1)
    res['gen_images_s'] = images[:, :, :, ::8, ::8]
    save('artv_tiny', meta=dict(seed=19, vae_seed=11, torch_seed=5), manifest=man, **res)


def case_mask_predict():
    m, man = _build_bert(0, False, 17)
    text = synth_tokens('text', (2, 16), 49408, 17, low=1)
    text[0, 11:] = 0
    text[1, 5:] = 0
    m.eval()
    res = {}
    for tag, steps, dyn, B in (('s4', 4, False, 1), ('s8dynB2', 8, True, 2)):
        cfg = dict(MP_CONFIG, B=B)
        seed_all(31)
        images, _, img_seq = m.generate_images(text, mask_predict_steps=steps, mp_config=cfg, dynamic=dyn)
        res[tag + '_img_seq'] = img_seq
        res[tag + '_images_s'] = images[:, :, :, ::8, ::8]
        res[tag + '_images_sum'] = images.double().sum().view(1)
    save('mask_predict', meta=dict(seed=17, vae_seed=11, torch_seed=31, mp_config=MP_CONFIG), **res)


def case_frontend():
    """The stochastic front-end's building blocks, run from the REFERENCE's own functions (dalle_bert.py:93-238, 992-1029):
    outputs together with the decisions its generators drew, recovered by replaying the same draws from the same generator
    states (the replay below follows the reference's draw order line by line; it decides nothing about the arithmetic)."""
    import mmvid_pytorch.dalle_bert as db
    res, meta = {}, {}
    S = 32
    # -- warp_with_affine (168-202): theta from the 4 uniform_ draws, replayed
    frame = synth_input('frame', (3, S, S), 41, 'uniform')
    res['frame'] = frame
    aff_p, aff_o = [], []
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        out = db.warp_with_affine(frame, 30, 0.1, 0.1)
        torch.manual_seed(seed)
        pa = torch.FloatTensor(4)
        ang = np.pi * 30 / 180.
        pa[0].uniform_(-ang, ang), pa[1].uniform_(-0.1, 0.1), pa[2].uniform_(-0.1, 0.1), pa[3].uniform_(0.9, 1.1)
        aff_p.append(pa.clone()), aff_o.append(out[0])
    res['affine_params'], res['affine_out'] = torch.stack(aff_p), torch.stack(aff_o)  # (angle, t1, t2, scale)
    # -- warp_with_color (124-135): c_shift = torch.rand(1) - 0.5, num = random.randint(0, 3)
    col_p, col_o = [], []
    for seed in (1, 2, 3, 4, 5, 6):
        seed_all(seed)
        out = db.warp_with_color(frame)
        seed_all(seed)
        shift = float(torch.rand(1) - 0.5)
        num = random.randint(0, 3)
        col_p.append([shift, num]), col_o.append(out[0])
    res['color_params'], res['color_out'] = torch.tensor(col_p, dtype=torch.float64), torch.stack(col_o)
    # -- warp_video_with_color (140-158; visual_aug_mode == 'motion_color'): per sample one shift for every frame
    video = synth_input('video', (4, 3, 3, 16, 16), 42, 'uniform')
    seed_all(9)
    vout = db.warp_video_with_color(video)
    seed_all(9)
    vp = []
    for _ in range(video.shape[0]):
        shift = float(torch.rand(1) - 0.5)
        vp.append([shift, random.randint(0, 3)])
    res['video'], res['video_color_params'], res['video_color_out'] = video, torch.tensor(vp, dtype=torch.float64), vout
    # -- swap (110-122), even batch: halves exchanged
    ctl = synth_input('ctl', (4, 5, 8), 43, 'normal')
    res['swap_in'], res['swap_out'] = ctl, db.swap(ctl, 0)
    # -- warp (204-238): B = 4 samples x 8 seeds so every strategy occurs; decisions replayed in the reference's draw order
    x = synth_input('clip', (4, 6, 3, 16, 16), 44, 'uniform')  # t = 6: randperm takes the torch.randperm branch (n >= 6)
    res['clip'] = x
    prob = [0.25, 0.25, 0.25, 0.25]
    rows, outs = [], []
    for seed in range(8):
        seed_all(100 + seed)
        y = db.warp(x, prob)
        seed_all(100 + seed)
        b, t = x.shape[:2]
        for i in range(b):  # (mode, j1, src_b, src_t, chan, shift, angle, t1, t2, scale, perm[6])
            rec = [0.] * 16
            strategy = int(np.random.choice(range(4), p=prob))
            rec[0] = strategy
            if strategy == 0:
                i_ = int(np.random.choice(list(set(range(b)) - {i})))
                j1, j2 = random.randint(0, t - 1), random.randint(0, t - 1)
                rec[1], rec[2], rec[3] = j1, i_, j2
            elif strategy == 1:
                perm_ord = torch.tensor(range(t))
                while True:
                    perm = torch.randperm(t)
                    if (perm != perm_ord).any():
                        break
                rec[10:16] = [float(v) for v in perm]
            elif strategy == 2:
                rec[1] = random.randint(0, t - 1)
                rec[5] = float(torch.rand(1) - 0.5)
                rec[4] = random.randint(0, 3)
            else:
                rec[1] = random.randint(0, t - 1)
                pa = torch.FloatTensor(4)
                ang = np.pi * 30 / 180.
                pa[0].uniform_(-ang, ang), pa[1].uniform_(-0.1, 0.1), pa[2].uniform_(-0.1, 0.1), pa[3].uniform_(0.9, 1.1)
                rec[6:10] = [float(v) for v in pa]
            rows.append(rec)
        outs.append(y)
    res['warp_decisions'] = torch.tensor(rows, dtype=torch.float64).view(8, 4, 16)
    res['warp_out'] = torch.stack(outs)
    # -- MSM masking loop (992-1029) inside BERT.forward: strategies / Bernoulli p from the numpy stream (replayed), boxes
    #    recorded from the RandomErasing stand-in, the Bernoulli field and the final mask1 from the forward itself
    m, man = _build_bert(0, False, 17, num_targets=4)
    B, T, TL = 8, 4, 16
    text = synth_tokens('text', (B, TL), 49408, 17, low=1)
    tok = synth_tokens('tok', (B, T * 16), 256, 17)
    boxes, bern = [], []
    get_params = ref_stubs.RandomErasing.get_params

    def rec_params(img, scale, ratio, value=None):
        r = get_params(img, scale, ratio, value)
        boxes.append([int(r[0]), int(r[1]), int(r[2]), int(r[3])])
        return r

    bern_orig = torch.bernoulli

    def rec_bern(p, *a, **k):
        r = bern_orig(p, *a, **k)
        bern.append(r.clone())
        return r

    cap = []
    h = m.image_emb.register_forward_hook(lambda mod, i, o: cap.append(i[0].clone()))
    m.train()
    msm_rows = []
    strat_prob = [0.3, 0.1, 0.3, 0.3]
    for case, (seed, pc_prob) in enumerate(((7, 0.0), (8, 0.0), (9, 0.6))):
        boxes.clear(), bern.clear(), cap.clear()
        ref_stubs.RandomErasing.get_params = staticmethod(rec_params)
        torch.bernoulli = rec_bern
        seed_all(seed)
        np_state, py_state = np.random.get_state(), random.getstate()
        with torch.no_grad():
            m(text, target=tok, return_loss=True, rel=False, vid=False, msm_strategy_prob=np.array(strat_prob),
              msm_bernoulli_prob=[0.2, 0.5], pc_prob=pc_prob)
        ref_stubs.RandomErasing.get_params = staticmethod(get_params)
        torch.bernoulli = bern_orig
        mask1 = cap[0] != m.image_token_lut['[MASK]']
        # replay the numpy / python streams in the loop's order
        np.random.set_state(np_state), random.setstate(py_state)
        bi, ki = 0, 0
        for i in range(B):
            which = int(np.random.choice([1, 2, 3, 4], p=strat_prob))
            p, box, field, keep = 0.0, [0, 0, 0, 0], torch.zeros(T * 16), [0.] * T
            if which == 1:
                p = float(np.random.uniform(0.2, 0.5))
                field = bern[ki]
                ki += 1
            elif which >= 3:
                box = boxes[bi]
                bi += 1
            if pc_prob > 0 and random.random() < pc_prob:
                t_overlap = random.randint(1, T // 2)
                for tt in random.sample(range(T), t_overlap):
                    keep[tt] = 1.
            msm_rows.append(dict(case=case, strategy=which, p=p, box=box, keep_frames=keep, bern=field, mask1=mask1[i]))
        assert bi == len(boxes) and ki == len(bern)
    h.remove()
    res['msm_strategy'] = torch.tensor([r['strategy'] for r in msm_rows])
    res['msm_p'] = torch.tensor([r['p'] for r in msm_rows], dtype=torch.float64)
    res['msm_box'] = torch.tensor([r['box'] for r in msm_rows])  # (i, j, h, w) of the erased rectangle on the f x f map
    res['msm_keep_frames'] = torch.tensor([r['keep_frames'] for r in msm_rows])
    res['msm_bernoulli'] = torch.stack([r['bern'] for r in msm_rows])  # the reference's torch.bernoulli draw (strategy 1)
    res['msm_mask1'] = torch.stack([r['mask1'] for r in msm_rows])
    meta.update(T=T, fmap=4, strategies_seen=sorted(set(int(v) for v in res['msm_strategy'])),
                warp_modes_seen=sorted(set(int(v) for v in res['warp_decisions'][..., 0].flatten())))
    assert meta['strategies_seen'] == [1, 2, 3, 4] and meta['warp_modes_seen'] == [0, 1, 2, 3], meta
    save('frontend', meta=meta, **res)


def case_mask_predict_race():
    """The reference's mask_predict (dalle_bert.py:514-714) with torch.multinomial replaced by the exponential race it
    implements (q ~ Exp(1) per category; with replacement: argmax p / q; without: the k largest), the variates recorded:
    every draw's inputs and decisions, the state after every step, the final tokens.  oracle/sampling.py and the HIP
    sampler must take the same decisions from the same variates."""
    m, man = _build_bert(0, False, 17)
    text = synth_tokens('text', (2, 16), 49408, 17, low=1)
    text[0, 11:] = 0
    text[1, 5:] = 0
    m.eval()
    real_multinomial = torch.multinomial
    res = {}
    for tag, nvid, steps, dyn, Bm in (('a', 2, 4, False, 2), ('b', 1, 9, True, 1)):
        gen = torch.Generator().manual_seed(77)
        calls, logit_cap, itok_cap, zr, zv = [], [], [], [], []

        def race_multinomial(p, k, replacement=False):
            E = torch.empty(p.shape).exponential_(generator=gen)
            key = torch.where(p > 0, E / p, torch.full_like(p, float('inf')))
            if p.dim() == 2:
                assert k == 1
                idx = key.argmin(1, keepdim=True)
            else:
                if k <= 0 or k > int((p > 0).sum()):
                    raise RuntimeError('invalid multinomial draw (as torch.multinomial raises)')
                idx = torch.sort(key, stable=True)[1][:k]
            calls.append(dict(E=E, p=p.clone(), k=k, idx=idx.clone()))
            return idx

        hooks = [m.to_logits.register_forward_hook(lambda mod, i, o: logit_cap.append(o[0].clone())),
                 m.image_emb.register_forward_hook(lambda mod, i, o: itok_cap.append(i[0].clone())),
                 m.to_logits_rel.register_forward_hook(lambda mod, i, o: zr.append(o.reshape(-1).clone())),
                 m.to_logits_vid.register_forward_hook(lambda mod, i, o: zv.append(o.reshape(-1).clone()))]
        cfg = dict(MP_CONFIG, B=Bm)
        torch.multinomial = race_multinomial
        try:
            seed_all(31)
            with torch.no_grad():
                ctrl = m(text[:nvid], return_loss=False)
                seq, _ = m.mask_predict(ctrl, dynamic=dyn, steps=steps, mp_config=cfg)
        finally:
            torch.multinomial = real_multinomial
            for h in hooks:
                h.remove()
        tok_calls = [c for c in calls if c['p'].dim() == 2]
        keep_calls = [c for c in calls if c['p'].dim() == 1]
        assert len(tok_calls) == len(logit_cap)
        res[tag + '_final'] = seq
        res[tag + '_logits'] = torch.stack(logit_cap)  # one [TS, V] per tower pass, in call order
        res[tag + '_E_tok'] = torch.stack([c['E'] for c in tok_calls])
        res[tag + '_tok'] = torch.stack([c['idx'].view(-1) for c in tok_calls])
        res[tag + '_Y_tok'] = torch.stack([torch.gather(c['p'], 1, c['idx']).view(-1) for c in tok_calls])
        res[tag + '_E_keep'] = torch.stack([c['E'] for c in keep_calls])
        res[tag + '_Y_keep'] = torch.stack([c['p'] for c in keep_calls])  # the confidences the keep draw saw
        res[tag + '_k_keep'] = torch.tensor([c['k'] for c in keep_calls])
        keep = torch.zeros(len(keep_calls), seq.shape[1], dtype=torch.bool)
        for r, c in enumerate(keep_calls):
            keep[r, c['idx']] = True
        res[tag + '_keep'] = keep
        # image_emb inputs: [MASK] row (target_pos_emb setup), then per video tok_in and, per (step, candidate), I_tok BEFORE it
        res[tag + '_itok'] = torch.stack([t.view(-1) for t in itok_cap[1:]])
        res[tag + '_z_rel'], res[tag + '_z_vid'] = torch.cat(zr), torch.cat(zv)
        res[tag + '_tower_passes'] = torch.tensor([len(logit_cap)])
    save('mask_predict_race', meta=dict(seed=17, vae_seed=11, race_seed=77, mp_config=MP_CONFIG,
                                        cases=dict(a=dict(videos=2, steps=4, dynamic=False, B=2),
                                                   b=dict(videos=1, steps=9, dynamic=True, B=1))), **res)


CASES = dict(vq=case_vq, vqgan_tiny=case_vqgan_tiny, vqgan_full=case_vqgan_full, vqgan_full16=case_vqgan_full16,
             vqgan_full16_refinit=case_vqgan_full16_refinit, tower=case_tower, tower12=case_tower12,
             bert_tiny=case_bert_tiny, bert_tiny_visual=case_bert_tiny_visual, bert_negvc=case_bert_negvc, bert_negvc_visual=case_bert_negvc_visual, bert_flm=case_bert_flm,
             bert_flm_bottleneck=case_bert_flm_bottleneck, artv_tiny=case_artv_tiny,
             mask_predict=case_mask_predict, frontend=case_frontend, mask_predict_race=case_mask_predict_race)

if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    for c in (sys.argv[1:] or list(CASES)):
        print('==', c)
        CASES[c]()
