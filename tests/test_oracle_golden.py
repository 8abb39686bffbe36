"""The oracle (oracle/) against outputs of the reference itself (tests/golden/, made by
tools/make_golden.py).  CPU only.  Bars: token indices exact; fp32 tensors max|d|/max|ref| <= 1e-4."""
import random

import numpy as np
import pytest
import torch

from conftest import relerr, synth_model_sd
from oracle import artv, bert, tower, vqgan
from oracle.synth import synth_input, synth_state_dict, synth_tensor, synth_tokens
from oracle.vq import vq_argmin

TOL = 1e-4


@pytest.mark.parametrize('tag,n', [('sep', 1024), ('stress', 1024), ('small', 256)])
def test_vq_argmin_c_oracle_matches_reference(golden, tag, n):
    g = golden('vq')
    cb = (synth_input('cb_' + tag, (n, 256), 7, 'uniform') * 2 - 1) / n if tag == 'stress' else \
        synth_tensor('quantize.embedding.weight', (n, 256), 7)
    z = synth_input('z_' + tag, (512, 256), 7)
    idx, dmin = vq_argmin(z, cb)
    ref = g[tag + '_idx']
    mism = (idx != ref).nonzero().view(-1)
    # any disagreement must be a near-tie of the reference's own distances (<= 4 ulp at |d|)
    top2 = g[tag + '_top2_d']
    gap = top2[:, 1] - top2[:, 0]
    ulp = torch.tensor(np.spacing(top2[:, 0].abs().numpy()))
    assert (gap[mism] <= 4 * ulp[mism]).all()
    if tag != 'stress':
        assert len(mism) == 0
    assert torch.allclose(dmin, top2[:, 0], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('name', ['vqgan_tiny', 'vqgan_full'])
def test_vqgan_encode_decode(golden, name):
    g = golden(name)
    meta = g.meta
    sd = synth_state_dict(g.manifest, 11)
    s = meta['image_size']
    img = synth_input('img', (meta['n'], 3, s, s), 11, 'uniform')
    with torch.no_grad():
        z = vqgan.encode_z(sd, img, s)
        idx = vqgan.get_codebook_indices(sd, img, s)
        dec = vqgan.decode(sd, g['indices'], s)
    assert relerr(z, g['z_e']) <= TOL
    assert torch.equal(idx, g['indices'])
    assert relerr(dec, g['decoded']) <= TOL


@pytest.mark.parametrize('name', ['vqgan_full16', 'vqgan_full16_refinit'])
def test_vqgan_wide_index_goldens(golden, name):
    """Round 6: 16 full-size frames = 1,024 tokens per case, two weight seeds, the synthetic (well separated) codebook and the
    reference's own near-uniform initialisation: the oracle's indices, distances and z equal the reference's."""
    from conftest import wide_vqgan_case
    g = golden(name)
    sd, img = wide_vqgan_case(g)
    with torch.no_grad():
        idx = vqgan.get_codebook_indices(sd, img, 128)
        z = vqgan.encode_z(sd, img[:g.meta['z_frames']], 128)
    assert g['indices'].numel() >= 1024
    assert torch.equal(idx, g['indices'])
    assert relerr(z, g['z_e']) <= TOL
    # the C oracle on the reference's z rows reproduces index and both top-2 distances' winner
    zf = g['z_e'].permute(0, 2, 3, 1).reshape(-1, g['z_e'].shape[1]).contiguous()
    ci, cd = vq_argmin(zf, sd['model.quantize.embedding.weight'])
    n = zf.shape[0]
    assert torch.equal(ci, g['indices'].reshape(-1)[:n])
    assert torch.allclose(cd, g['top2_d'][:n, 0], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('tag,L,mt,idx', [('L51', 51, 'mask_prev', [17, 18]), ('L579', 579, 'mask_prev', [65, 66]),
                                          ('causal40', 40, 'causal', [])])
def test_tower_forward_backward(golden, tag, L, mt, idx):
    g = golden('tower')
    sd = synth_state_dict(g.manifest, 13)
    for v in sd.values():
        v.requires_grad_(True)
    x = synth_input('x_' + tag, (2, L, 768), 13).requires_grad_(True)
    gy = synth_input('g_' + tag, (2, L, 768), 13)
    y = tower.tower(sd, x, tower.build_attention_mask(L, mt, idx), 'transformer.')
    y.backward(gy)
    if L <= 64:
        assert relerr(y, g[tag + '_y']) <= TOL and relerr(x.grad, g[tag + '_dx']) <= TOL
    else:
        assert relerr(y[:, ::37, ::13], g[tag + '_y_s']) <= TOL
        assert relerr(x.grad[:, ::37, ::13], g[tag + '_dx_s']) <= TOL
    assert abs(y.double().norm().item() / g[tag + '_y_norm'].item() - 1) < 1e-5
    p = 'transformer.resblocks.'
    for nm, k in (('inw', p + '0.attn.in_proj_weight'), ('outw', p + '1.attn.out_proj.weight'),
                  ('fcw', p + '0.mlp.c_fc.weight'), ('pjw', p + '1.mlp.c_proj.weight')):
        assert relerr(sd[k].grad[::61, ::29], g[f'{tag}_d{nm}_s']) <= TOL
    for nm, k in (('inb', p + '0.attn.in_proj_bias'), ('ln1w', p + '0.ln_1.weight'), ('ln2b', p + '1.ln_2.bias'),
                  ('fcb', p + '1.mlp.c_fc.bias')):
        assert relerr(sd[k].grad, g[f'{tag}_d{nm}']) <= TOL


def test_tower_12_layers_at_training_length(golden):
    """The full-depth tower (12 layers, L = 579, restricted rows 65 / 66) against the reference's own forward + backward
    (tests/golden/tower12.npz, tools/make_golden.py::case_tower12): output / input-gradient slices, the rows around the
    restricted ones, norms, and parameter gradients of layers 0, 5 and 11."""
    g = golden('tower12')
    sd = synth_state_dict(g.manifest, 23)
    for v in sd.values():
        v.requires_grad_(True)
    L = g.meta['L']
    x = synth_input('x_t12', (2, L, 768), 23).requires_grad_(True)
    gy = synth_input('g_t12', (2, L, 768), 23)
    y = tower.tower(sd, x, tower.build_attention_mask(L, 'mask_prev', [65, 66]), 'transformer.')
    y.backward(gy)
    rows = [0, 64, 65, 66, 67, 578]
    assert relerr(y[:, ::37, ::13], g['y_s']) <= TOL and relerr(x.grad[:, ::37, ::13], g['dx_s']) <= TOL
    assert relerr(y[:, rows][..., ::7], g['y_rows']) <= TOL and relerr(x.grad[:, rows][..., ::7], g['dx_rows']) <= TOL
    assert abs(y.double().norm().item() / g['y_norm'].item() - 1) < 1e-5
    assert abs(x.grad.double().norm().item() / g['dx_norm'].item() - 1) < 1e-5
    p = 'transformer.resblocks.'
    for li in (0, 5, 11):
        for nm, k in (('inw', 'attn.in_proj_weight'), ('outw', 'attn.out_proj.weight'), ('fcw', 'mlp.c_fc.weight'),
                      ('pjw', 'mlp.c_proj.weight')):
            assert relerr(sd[f'{p}{li}.{k}'].grad[::61, ::29], g[f'l{li}_d{nm}_s']) <= TOL
        for nm, k in (('inb', 'attn.in_proj_bias'), ('ln1w', 'ln_1.weight'), ('ln2b', 'ln_2.bias'), ('fcb', 'mlp.c_fc.bias'),
                      ('pjb', 'mlp.c_proj.bias')):
            assert relerr(sd[f'{p}{li}.{k}'].grad, g[f'l{li}_d{nm}']) <= TOL


@pytest.mark.parametrize('name,nv', [('bert_tiny', 0), ('bert_tiny_visual', 1)])
def test_bert_losses_and_grads(golden, name, nv):
    g = golden(name)
    sd = synth_model_sd(g, 17)
    for k in sd:
        sd[k].requires_grad_(not k.startswith(('vae.', 'cvae.')))
    cfg = bert.Cfg(sd, 16, nv, 2, 64, use_cvae=nv > 0)
    text, frames = g['text'], g['frames']
    with torch.no_grad():
        tt = bert.get_image_tokens(sd, cfg, frames)
        wt = bert.get_image_tokens(sd, cfg, g['warped_frames'])
        vt = bert.get_image_tokens(sd, cfg, g['visual'], 'cvae') if nv else None
    assert torch.equal(tt, g['target_tok']) and torch.equal(wt, g['warp_tok'])
    if nv:
        assert torch.equal(vt, g['visual_tok'])
    r = bert.forward_losses(sd, cfg, text, tt, g['mask1'], wt, vt)
    assert relerr(r['control_emb'], g['control_emb']) <= TOL
    assert relerr(r['tokens_msm'], g['tokens_msm']) <= TOL
    assert relerr(r['out_msm'], g['out_msm']) <= TOL
    assert relerr(r['logits_msm'], g['logits_msm']) <= TOL
    losses = torch.stack([r['loss_msm'], r['loss_rel'], r['loss_vid']])
    assert torch.allclose(losses, g['losses'], rtol=1e-5)
    (7 * r['loss_msm'] + .5 * r['loss_rel'] + .5 * r['loss_vid']).backward()
    G = {k: v.grad for k, v in sd.items() if v.grad is not None}
    assert relerr(G['image_emb.weight'][::3, ::5], g['g_image_emb']) <= TOL
    assert relerr(G['to_logits.1.weight'][::4, ::6], g['g_to_logits_w']) <= TOL
    assert relerr(G['special_emb.weight'], g['g_special_emb']) <= TOL
    assert relerr(G['target_pos_emb.weights_0'].reshape(-1, 768), g['g_tpos0']) <= TOL
    assert relerr(G['transformer.transformer.resblocks.0.ln_1.weight'], g['g_ln1w']) <= TOL
    assert relerr(G['text_emb.weight'][g['g_text_emb_row_ids']][:, ::11], g['g_text_emb_rows']) <= TOL
    tn = torch.sqrt(sum((v.double()**2).sum() for v in G.values()))
    assert abs(tn.item() / g['g_total_norm'].item() - 1) < 1e-5


def test_bert_negvc_text_only(golden):
    """negvc=True without visuals: the reference's own run (tools/make_golden.py::case_bert_negvc) against the restatement."""
    g = golden('bert_negvc')
    sd = synth_model_sd(g, 17)
    for k in sd:
        sd[k].requires_grad_(not k.startswith(('vae.', 'cvae.')))
    cfg = bert.Cfg(sd, 16, 0, 2, 64)
    with torch.no_grad():
        tt = bert.get_image_tokens(sd, cfg, g['frames'])
        wt = bert.get_image_tokens(sd, cfg, g['warped_frames'])
    assert torch.equal(tt, g['target_tok']) and torch.equal(wt, g['warp_tok'])
    r = bert.forward_losses(sd, cfg, g['text'], tt, g['mask1'], wt, None, text_neg=g['text_neg'])
    assert relerr(r['tokens_rel'][:, :19], g['tokens_rel_ctrl']) <= TOL
    assert relerr(r['out_msm'][:, ::3, ::7], g['out_msm_s']) <= TOL
    assert relerr(r['out_rel'][:, ::3, ::7], g['out_rel_s']) <= TOL
    losses = torch.stack([r['loss_msm'], r['loss_rel'], r['loss_vid']])
    assert torch.allclose(losses, g['losses'], rtol=1e-5)
    (7 * r['loss_msm'] + .5 * r['loss_rel'] + .5 * r['loss_vid']).backward()
    G = {k: v.grad for k, v in sd.items() if v.grad is not None}
    assert relerr(G['special_emb.weight'], g['g_special_emb']) <= TOL
    assert relerr(G['text_pos_emb.weight'][:, ::5], g['g_text_pos']) <= TOL
    assert relerr(G['to_logits_rel.1.weight'], g['g_relw']) <= TOL
    assert relerr(G['text_emb.weight'][g['g_text_emb_row_ids']][:, ::11], g['g_text_emb_rows']) <= TOL
    tn = torch.sqrt(sum((v.double()**2).sum() for v in G.values()))
    assert abs(tn.item() / g['g_total_norm'].item() - 1) < 1e-5


def test_bert_negvc_with_visuals(golden):
    """negvc=True together with a visual control: the reference's own run (tools/make_golden.py::case_bert_negvc_visual).  Its REL-negative
    pass has no visual segment -- 51 positions against 67 -- and `visual_neg` is ignored."""
    g = golden('bert_negvc_visual')
    assert g.meta['pass_lengths'] == [67, 51, 67]
    sd = synth_model_sd(g, 17)
    for k in sd:
        sd[k].requires_grad_(not k.startswith(('vae.', 'cvae.')))
    cfg = bert.Cfg(sd, 16, 1, 2, 64, use_cvae=True)
    with torch.no_grad():
        tt = bert.get_image_tokens(sd, cfg, g['frames'])
        wt = bert.get_image_tokens(sd, cfg, g['warped_frames'])
        vt = bert.get_image_tokens(sd, cfg, g['visual'], 'cvae')
    assert torch.equal(tt, g['target_tok']) and torch.equal(wt, g['warp_tok']) and torch.equal(vt, g['visual_tok'])
    r = bert.forward_losses(sd, cfg, g['text'], tt, g['mask1'], wt, vt, text_neg=g['text_neg'])
    assert r['tokens_rel'].shape[1] == 51
    assert relerr(r['tokens_rel'][:, :, ::5], g['tokens_rel']) <= TOL
    assert relerr(r['out_msm'][:, ::3, ::7], g['out_msm_s']) <= TOL
    assert relerr(r['out_rel'][:, :, ::7], g['out_rel']) <= TOL
    assert relerr(r['out_vid'][:, ::3, ::7], g['out_vid_s']) <= TOL
    losses = torch.stack([r['loss_msm'], r['loss_rel'], r['loss_vid']])
    assert torch.allclose(losses, g['losses'], rtol=1e-5)
    (7 * r['loss_msm'] + .5 * r['loss_rel'] + .5 * r['loss_vid']).backward()
    G = {k: v.grad for k, v in sd.items() if v.grad is not None}
    for key, name in (('special_emb.weight', 'g_special_emb'), ('to_logits_rel.1.weight', 'g_relw'),
                      ('transformer.transformer.resblocks.1.mlp.c_fc.bias', 'g_fcb')):
        assert relerr(G[key], g[name]) <= TOL
    assert relerr(G['text_pos_emb.weight'][:, ::5], g['g_text_pos']) <= TOL
    assert relerr(G['visual_emb.weight'][::3, ::5], g['g_visual_emb']) <= TOL
    assert relerr(G['text_emb.weight'][g['g_text_emb_row_ids']][:, ::11], g['g_text_emb_rows']) <= TOL
    tn = torch.sqrt(sum((v.double()**2).sum() for v in G.values()))
    assert abs(tn.item() / g['g_total_norm'].item() - 1) < 1e-5


@pytest.mark.parametrize('name', ['bert_flm', 'bert_flm_bottleneck'])
def test_bert_fixed_language_model(golden, name):
    """dalle_bert.py:307-322, 924-925: the text is one mapped sentence feature (single Linear / LayerNorm-Linear bottleneck)."""
    g = golden(name)
    sd = synth_model_sd(g, 23)
    for k in sd:
        sd[k].requires_grad_(not k.startswith('vae.'))
    assert 'text_emb.weight' not in sd and any(k.startswith('text_feature_mapping.') for k in sd)
    cfg = bert.Cfg(sd, 1, 0, 2, 64)
    assert cfg.total_seq_len == 1 + 1 + 2 + 2 * 16
    feat = g['text_feat']
    with torch.no_grad():
        tt = bert.get_image_tokens(sd, cfg, g['frames'])
    assert torch.equal(tt, g['target_tok'])
    r = bert.forward_losses(sd, cfg, feat, tt, g['mask1'], g['warp_tok'], None)
    assert relerr(r['control_emb'], g['control_emb']) <= TOL
    assert relerr(r['out_msm'][:, ::3, ::7], g['out_msm_s']) <= TOL
    assert relerr(r['out_rel'][:, ::3, ::7], g['out_rel_s']) <= TOL and relerr(r['out_vid'][:, ::3, ::7], g['out_vid_s']) <= TOL
    losses = torch.stack([r['loss_msm'], r['loss_rel'], r['loss_vid']])
    assert torch.allclose(losses, g['losses'], rtol=1e-5)
    (7 * r['loss_msm'] + .5 * r['loss_rel'] + .5 * r['loss_vid']).backward()
    G = {k: v.grad for k, v in sd.items() if v.grad is not None}
    n = 0
    for k, v in G.items():
        if k.startswith('text_feature_mapping.'):
            assert relerr(v if v.dim() == 1 else v[::4, ::8], g['g_' + k]) <= TOL, k
            n += 1
    assert n == (2 if name == 'bert_flm' else 10)
    assert relerr(G['image_emb.weight'][::3, ::5], g['g_image_emb']) <= TOL
    tn = torch.sqrt(sum((v.double()**2).sum() for v in G.values()))
    assert abs(tn.item() / g['g_total_norm'].item() - 1) < 1e-5


def test_mask_predict_trajectory(golden):
    g, gb = golden('mask_predict'), golden('bert_tiny')
    sd = synth_model_sd(gb, 17)
    cfg = bert.Cfg(sd, 16, 0, 2, 64)
    text = synth_tokens('text', (2, 16), 49408, 17, low=1)
    text[0, 11:] = 0
    text[1, 5:] = 0
    for tag, steps, dyn, B in (('s4', 4, False, 1), ('s8dynB2', 8, True, 2)):
        random.seed(31), np.random.seed(31), torch.manual_seed(31)
        imgs, seq = bert.generate_images(sd, cfg, text, steps, dict(g.meta['mp_config'], B=B), dyn)
        assert torch.equal(seq, g[tag + '_img_seq'])
        assert relerr(imgs[:, :, :, ::8, ::8], g[tag + '_images_s']) <= TOL


def test_artv_logits_loss_and_sampling(golden):
    g = golden('artv_tiny')
    sd = synth_model_sd(g, 19)
    cfg = artv.Cfg(sd, 16, 1, 2, 64)
    text = g['text']
    with torch.no_grad():
        tt = vqgan.get_codebook_indices(sd, g['frames'].reshape(-1, 3, 64, 64), 64, 'vae.model.').view(2, -1)
        vt = vqgan.get_codebook_indices(sd, g['visual'].reshape(-1, 3, 64, 64), 64, 'vae.model.').view(2, -1)
        assert torch.equal(tt, g['target_tok'])
        assert abs(artv.forward(sd, cfg, text, vt, tt, True).item() - g['loss'].item()) < 1e-5
        assert abs(artv.forward(sd, cfg, text, None, tt, True).item() - g['loss_novisual'].item()) < 1e-5
        for k in (0, 5, 31):
            last = artv.forward(sd, cfg, text, vt, tt[:, :k])[:, -1]
            assert relerr(last[:, cfg.num_control_tokens:], g[f'logits_k{k}_img']) <= TOL
            assert torch.equal(last.argmax(-1), g[f'logits_k{k}_argmax'])
        random.seed(5), np.random.seed(5), torch.manual_seed(5)
        imgs, _ = artv.generate_images(sd, cfg, text[:1], vt[:1])
        assert relerr(imgs[:, :, :, ::8, ::8], g['gen_images_s']) <= TOL


def test_race_sampler_matches_multinomial_distributions():
    """oracle/sampling.py (randomness injected as Exp(1) race variates) draws from the distributions torch.multinomial
    draws from -- the form oracle/bert.py::mask_predict uses and the reference golden trajectory pins."""
    from oracle import sampling as S
    rng = np.random.RandomState(0)
    p = np.array([0.45, 0.25, 0.15, 0.1, 0.05], np.float32)
    n = 40000
    tok, Y, P = S.token_race(np.log(p)[None].repeat(n, 0), rng.exponential(size=(n, 5)).astype(np.float32))
    freq = np.bincount(tok, minlength=5) / n
    assert np.abs(freq - p).max() < 4 * np.sqrt(0.25 / n)
    assert np.allclose(Y, p[tok], rtol=1e-5)
    w = np.array([0.4, 0.3, 0.2, 0.1, 0.0, 0.25], np.float32)
    inc = np.zeros(6)
    for _ in range(6000):
        inc += S.keep_race(w, rng.exponential(size=6).astype(np.float32), None, 3)
    ref = torch.zeros(6)
    torch.manual_seed(0)
    for _ in range(6000):
        ref[torch.multinomial(torch.from_numpy(w), 3, replacement=False)] += 1
    assert np.abs(inc / 6000 - ref.numpy() / 6000).max() < 0.03 and inc[4] == 0
    # the reference's except-branch: an impossible request degrades to ONE kept position
    assert S.keep_race(w, rng.exponential(size=6).astype(np.float32), None, 0).sum() == 1
    assert S.keep_race(w, rng.exponential(size=6).astype(np.float32), None, 6).sum() == 1
    pres = np.array([1, 0, 0, 0, 0, 0], bool)
    k = S.keep_race(w, rng.exponential(size=6).astype(np.float32), pres, 2)
    assert k[0] and k.sum() == 3


def _warp_params(dec):
    """tests/golden/frontend.npz `warp_decisions` rows -> oracle.frontend.apply_warp parameter dicts."""
    from oracle.frontend import affine_theta
    out = []
    for r in dec.tolist():
        out.append(dict(mode=int(r[0]), j1=int(r[1]), src_b=int(r[2]), src_t=int(r[3]), chan=int(r[4]), shift=float(np.float32(r[5])),
                        theta=affine_theta(*[float(np.float32(v)) for v in r[6:10]]), perm=[int(v) for v in r[10:16]]))
    return out


def test_frontend_oracle_pinned_to_reference_warps(golden):
    """H6 / N2 pin: oracle/frontend.py against the reference's OWN warp_with_affine / warp_with_color /
    warp_video_with_color / swap / warp (dalle_bert.py:93-238), given the decisions its generators drew."""
    from oracle import frontend as F
    g = golden('frontend')
    frame = g['frame']
    for p, ref in zip(g['affine_params'], g['affine_out']):
        got = F.affine_warp(frame, F.affine_theta(*[float(v) for v in p]))
        assert relerr(got, ref) <= 1e-6
    for (shift, num), ref in zip(g['color_params'].tolist(), g['color_out']):
        assert torch.equal(F.color_shift(frame, float(np.float32(shift)), int(num)), ref)
    vp = [(float(np.float32(a)), int(b)) for a, b in g['video_color_params'].tolist()]
    assert torch.equal(F.video_color_shift(g['video'], vp), g['video_color_out'])
    assert torch.equal(F.swap_halves(g['swap_in']), g['swap_out'])
    x = g['clip']
    seen = set()
    for dec, ref in zip(g['warp_decisions'], g['warp_out']):
        params = _warp_params(dec)
        seen |= {p['mode'] for p in params}
        got = F.apply_warp(x, params)
        for b, p in enumerate(params):
            if p['mode'] == 3:
                assert relerr(got[b], ref[b]) <= 1e-6
            else:
                assert torch.equal(got[b], ref[b]), (b, p['mode'])
    assert seen == {0, 1, 2, 3}


def test_frontend_oracle_pinned_to_reference_msm_loop(golden):
    """The MSM masking loop of BERT.forward (dalle_bert.py:992-1029) run by the reference: build_msm_mask applied to the
    recorded decisions (strategy, Bernoulli field, RandomErasing box, preserved frames) gives the reference's mask1."""
    from oracle.frontend import build_msm_mask
    g = golden('frontend')
    T, f = g.meta['T'], g.meta['fmap']
    strat, box, keep, bern, ref = g['msm_strategy'], g['msm_box'], g['msm_keep_frames'], g['msm_bernoulli'], g['msm_mask1']
    assert sorted(set(strat.tolist())) == [1, 2, 3, 4] and keep.sum() > 0
    for i in range(len(strat)):
        kf = [t for t in range(T) if keep[i, t] > 0]
        m, nfm = build_msm_mask(int(strat[i]), T, f, bern[i].numpy(), tuple(box[i].tolist()), kf)
        assert np.array_equal(m, ref[i].numpy()), (i, int(strat[i]))
        assert nfm == (0.0 if int(strat[i]) == 2 else 1.0)


def test_race_sampler_pinned_to_reference_mask_predict(golden):
    """oracle/sampling.py against the reference's mask_predict (dalle_bert.py:514-714) run with torch.multinomial replaced by
    the exponential race on recorded variates (tests/golden/mask_predict_race.npz): every token draw, keep set, candidate
    update / choice and the dynamic stop, decision for decision."""
    from oracle import sampling as S
    g = golden('mask_predict_race')
    for tag, c in g.meta['cases'].items():
        nv, steps, dyn, Bm = c['videos'], c['steps'], c['dynamic'], c['B']
        logits, Et, tok, Yt = g[tag + '_logits'], g[tag + '_E_tok'], g[tag + '_tok'], g[tag + '_Y_tok']
        Ek, Yk, kk, keep = g[tag + '_E_keep'], g[tag + '_Y_keep'], g[tag + '_k_keep'], g[tag + '_keep']
        itok, zr, zv, final = g[tag + '_itok'], g[tag + '_z_rel'], g[tag + '_z_vid'], g[tag + '_final']
        TS = final.shape[1]
        # every token draw: first argmin of E / P and its probability
        for r in range(logits.shape[0]):
            t_, Y_, _ = S.token_race(logits[r], Et[r])
            assert np.array_equal(t_, tok[r].numpy()), (tag, r)
            assert np.allclose(Y_, Yt[r].numpy(), rtol=2e-6, atol=0)
        ti = ki = ii = 0
        for v in range(nv):
            ii += 1  # tok_in of this video
            Y, I_tok = Yt[ti].numpy(), tok[ti].numpy()
            ti += 1
            Smax, tmax, Imax, stopped = 0.0, 0, None, False
            for t in range(1, steps):
                masks = []
                for j in range(Bm):
                    assert np.allclose(Yk[ki].numpy(), Y, rtol=0, atol=0), 'the confidences the reference drew the keep set from'
                    assert np.array_equal(itok[ii].numpy(), I_tok)
                    k = S.keep_race(Y, Ek[ki], None, int(kk[ki]))
                    assert np.array_equal(k, keep[ki].numpy()), (tag, v, t, j)
                    masks.append(k)
                    ki += 1
                    ii += 1
                Ynew = np.stack([Yt[ti + j].numpy() for j in range(Bm)])
                Inew = np.stack([tok[ti + j].numpy() for j in range(Bm)])
                Y, I_tok, Sc, jmax = S.update(Y, I_tok, np.stack(masks), Ynew, Inew, zr[ti - v - 1:ti - v - 1 + Bm], zv[ti - v - 1:ti - v - 1 + Bm])
                ti += Bm
                if dyn:
                    Smax, tmax, took, stop = S.dynamic_stop(float(Sc[jmax]), t, Smax, tmax)
                    if took:
                        Imax = I_tok
                    if stop:
                        stopped = True
                        break
                else:
                    Imax = I_tok
            assert np.array_equal(Imax, final[v].numpy()), (tag, v)
        assert ti == logits.shape[0] and ki == Ek.shape[0], (tag, ti, ki)
