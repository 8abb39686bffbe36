#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference) on CPU.

Build-container only: the reference is imported (never copied) through tools/ref_stubs.py;
weights and inputs are the deterministic synthetic ones of oracle/synth.py, so a fixture
holds seeds, the reference's state_dict manifest (key -> shape: pins the checkpoint
layout), and the reference's outputs.  Run:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [case ...]

Cases: vq vqgan_tiny vqgan_full tower bert_tiny bert_tiny_visual artv_tiny mask_predict
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


def _build_bert(num_visuals, use_cvae, seed, text_seq_len=16, num_targets=2):
    from mmvid_pytorch.dalle_bert import BERT
    ref_stubs.CLIP_STATE['sd'] = clip_state(2)
    vae, _ = build_vae(True, 11)
    cvae = build_vae(True, 12)[0] if use_cvae else None
    m = BERT(dim=768, vae=vae, cvae=cvae, num_text_tokens=49408, text_seq_len=text_seq_len,
             which_transformer='openai_clip_visual', num_visuals=num_visuals, num_targets=num_targets,
             openai_clip_path='x')
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


CASES = dict(vq=case_vq, vqgan_tiny=case_vqgan_tiny, vqgan_full=case_vqgan_full, tower=case_tower,
             bert_tiny=case_bert_tiny, bert_tiny_visual=case_bert_tiny_visual, artv_tiny=case_artv_tiny,
             mask_predict=case_mask_predict)

if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    for c in (sys.argv[1:] or list(CASES)):
        print('==', c)
        CASES[c]()
