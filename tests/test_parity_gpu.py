"""Round-2 parity tests on a real MI355X: exact token indices through the strict (fp32) VQGAN path, injected-randomness
sampler trajectories (SURVEY 8c vi), the device front-end's distributions, the fused heads, and the boundary closures
(decode_train, optimiser state, shadows)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr, synth_model_sd
from test_host_logic import tiny_bert, tiny_vae
from test_models_gpu import DEV, _with_tokens, close, load_synth

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------- strict VQGAN: exact indices
@pytest.mark.parametrize('name,tiny', [('vqgan_tiny', True), ('vqgan_full', False)])
def test_strict_encoder_indices_equal_reference(golden, name, tiny):
    """north star: "token-index bit-exact vs reference".  vae.strict = True runs every convolution / matmul as an fp32
    fmaf chain (csrc/strict.hip); the indices must equal the reference's, z must agree to fp32 round-off."""
    from mmvid_amd.vae import VQGanVAE1024
    from oracle.synth import synth_input
    g = golden(name)
    s = g.meta['image_size']
    vae = tiny_vae() if tiny else VQGanVAE1024(None, 128)
    vae.image_size = s
    load_synth(vae, g, 11)
    vae.strict = True
    img = synth_input('img', (g.meta['n'], 3, s, s), 11, 'uniform').to(DEV)
    z = vae.encode_z(img)
    close(z, g['z_e'].permute(0, 2, 3, 1), 2e-5, f'{name} strict z_e')
    idx = vae.get_codebook_indices(img).cpu()
    assert torch.equal(idx, g['indices']), f'{name}: {(idx != g["indices"]).sum().item()} of {idx.numel()} indices differ'
    dec = vae.decode(g['indices'].to(DEV))
    close(dec, g['decoded'], 2e-5, f'{name} strict decode')
    # batch composition must not matter, and the default path must be untouched by the switch
    assert torch.equal(vae.get_codebook_indices(img[:1]).cpu(), idx[:1])
    vae.strict = False
    assert vae.get_codebook_indices(img).shape == idx.shape


@pytest.mark.parametrize('name,nv,cvae', [('bert_tiny', 0, False), ('bert_tiny_visual', 1, True)])
def test_strict_tokens_of_bert_goldens(golden, name, nv, cvae):
    g = golden(name)
    m = load_synth(tiny_bert(nv, cvae), g, 17)
    m.vae.strict = True
    if m.cvae is not None:
        m.cvae.strict = True
    assert torch.equal(m.get_image_tokens(g['frames'].to(DEV)).cpu(), g['target_tok'])
    assert torch.equal(m.get_image_tokens(g['warped_frames'].to(DEV)).cpu(), g['warp_tok'])
    if nv:
        assert torch.equal(m.get_image_tokens(g['visual'].to(DEV), which_vae='cvae').cpu(), g['visual_tok'])


def test_token_helpers_recon_codebook_emb_masks(golden):
    """BERT.recon_images / get_codebook_emb / decode_images / decode_masks (dalle_bert.py:503-512, 754-778) on the HIP path,
    against the CPU oracle's VQGAN (strict mode: the same tokens, pixels within the fp32 tolerance) and plain indexing."""
    from oracle import vqgan
    g = golden('bert_tiny_visual')
    m = load_synth(tiny_bert(1, True), g, 17).eval()
    sd = synth_model_sd(g, 17)
    frames = g['frames'].to(DEV)  # [B, T, 3, 64, 64]
    B, T = frames.shape[:2]
    for which, prefix in (('vae', 'vae.model.'), ('cvae', 'cvae.model.')):
        vae = m.vae if which == 'vae' else m.cvae
        vae.strict = True
        idx_o = vqgan.get_codebook_indices(sd, g['frames'].reshape(-1, 3, 64, 64), 64, prefix)
        rec_o = vqgan.decode(sd, idx_o, 64, prefix)
        rec = m.recon_images(frames, which_vae=which)
        assert rec.shape == (B * T, 3, 64, 64)
        close(rec, rec_o, 2e-5, f'recon_images ({which}) vs oracle encode -> decode')
        code, emb = m.get_codebook_emb(frames, which_vae=which)
        assert code.shape == (B, T, 16) and torch.equal(code.cpu().view(B * T, -1), idx_o)
        assert torch.equal(emb, m.image_emb.weight.detach()[code])
        vae.strict = False
    tok = m.get_image_tokens(frames)
    close(m.decode_images(tok), m.vae.decode(tok.view(B * T, -1)), 0.0, 'decode_images == vae.decode of the reshaped tokens')
    mask = (torch.rand(B, T * 16, device=DEV) < 0.4).float()
    red = m.decode_masks(mask)
    assert red.shape == (B * T, 3, 64, 64) and float(red[:, 1:].abs().max()) == 0.0
    up = mask.view(B * T, 1, 4, 4).repeat_interleave(16, 2).repeat_interleave(16, 3)
    assert torch.equal(red[:, :1], up)


def test_strict_tokens_of_artv_golden(golden):
    g = golden('artv_tiny')
    vae = tiny_vae()
    sd = {k[4:]: v for k, v in synth_model_sd(g, 19).items() if k.startswith('vae.')}
    vae.load_state_dict(sd)
    vae.to(DEV).strict = True
    tok = vae.get_codebook_indices(g['frames'].reshape(-1, 3, 64, 64).to(DEV)).view(2, -1).cpu()
    assert torch.equal(tok, g['target_tok'])


def test_default_encoder_gap_histogram(golden):
    """The bf16 encoder: every index that differs from the reference must be a near-tie of the reference's own top-2
    distances; the histogram of those gaps is reported (no mismatch allowance)."""
    from mmvid_amd.vae import VQGanVAE1024
    from oracle.synth import synth_input
    g = golden('vqgan_full')
    vae = VQGanVAE1024(None, 128)
    vae.image_size = 128
    load_synth(vae, g, 11)
    img = synth_input('img', (g.meta['n'], 3, 128, 128), 11, 'uniform').to(DEV)
    idx = vae.get_codebook_indices(img).cpu()
    z = vae.encode_z(img).cpu()
    zerr = (z - g['z_e'].permute(0, 2, 3, 1)).abs().max().item()
    gap = (g['top2_d'][:, 1] - g['top2_d'][:, 0]).view_as(idx)
    mism = idx != g['indices']
    edges = [0, 1e-3, 1e-2, 3e-2, 1e-1, 3e-1, 1e9]
    hist_all = np.histogram(gap.numpy(), edges)[0].tolist()
    hist_mis = np.histogram(gap[mism].numpy(), edges)[0].tolist()
    print(f'bf16 encoder: {int(mism.sum())}/{idx.numel()} indices differ; max |dz| {zerr:.3e}; top-2 gap histogram '
          f'(edges {edges[:-1]}): all {hist_all}, mismatching {hist_mis}')
    assert (gap[mism] < 64 * zerr + 1e-3).all()


def test_decode_train_equals_decode_for_one_hot(golden):
    """vae.py:58-68: probs @ codebook -> decoder.  A one-hot probs tensor must reproduce decode(indices)."""
    g = golden('vqgan_tiny')
    vae = load_synth(tiny_vae(), g, 11)
    idx = g['indices'].to(DEV)
    probs = F.one_hot(idx, 256).float()
    a, b = vae.decode_train(probs), vae.decode(idx)
    close(a, b, 1e-6, 'decode_train(one-hot) vs decode')
    soft = torch.softmax(torch.randn(2, 16, 256, device=DEV), -1)
    out = vae.decode_train(soft)
    assert out.shape == b.shape and out.min() >= 0 and out.max() <= 1
    vae.strict = True
    close(vae.decode_train(probs), g['decoded'], 2e-5, 'strict decode_train vs reference decode')


# ------------------------------------------------------------------------------------------- samplers
def test_sample_race_kernel_matches_oracle():
    from mmvid_amd import ops
    from oracle import sampling as S
    torch.manual_seed(0)
    for R, V, temp in ((37, 1024, 0.0), (8, 256, 0.7), (5, 1088, 0.0)):
        logits = (torch.randn(R, V) * 3).to(DEV)
        E = torch.empty(R, V).exponential_().to(DEV)
        u = torch.rand(R, V).to(DEV) if temp else None
        tok, y = ops.sample_race(logits, E, u, temp)
        otok, oy, _ = S.token_race(logits, E, temp, u)
        assert np.array_equal(tok.cpu().numpy(), otok), (R, V, temp)
        assert np.allclose(y.cpu().numpy(), oy, rtol=2e-6, atol=1e-12)
    # the softmax divisor form (ART-V temperature) and a strided logits view
    logits = torch.randn(6, 2048, device=DEV)
    E = torch.empty(6, 1024, device=DEV).exponential_()
    tok, _ = ops.sample_race(logits[:, 512:1536], E, None, 0.0, logit_div=0.5, want_y=False)
    otok, _, _ = S.token_race(logits[:, 512:1536], E, logit_div=0.5)
    assert np.array_equal(tok.cpu().numpy(), otok)


def test_select_keep_kernel_matches_oracle():
    from mmvid_amd import ops
    from oracle import sampling as S
    torch.manual_seed(1)
    b, Bm, TS = 3, 2, 512
    Y = torch.rand(b, TS).to(DEV)
    Y[0, 5:40] = 0.0  # zero weights can never be drawn
    E = torch.empty(b, Bm, TS).exponential_().to(DEV)
    pres = (torch.arange(TS) < 64).to(torch.uint8).to(DEV)
    for preserve, k in ((None, 461), (None, 0), (pres, 100), (pres, 449), (None, 512), (None, 1)):
        m = ops.mp_select_keep(Y, E, preserve, k).cpu().numpy().astype(bool)
        for i in range(b):
            for j in range(Bm):
                ref = S.keep_race(Y[i], E[i, j], preserve, k)
                assert np.array_equal(m[i, j], ref), (k, i, j, m[i, j].sum(), ref.sum())


def _replay_trace(trace, fixed, b, Bm, dynamic):
    """Teacher-forced check of every decision of a mask-predict run against oracle/sampling.py."""
    from oracle import sampling as S
    t0 = trace[0]
    tok, Y, P = S.token_race(t0['logits'], t0['E_tok'])
    Iref, I0 = tok.reshape(b, -1), t0['I_tok'].cpu().numpy()
    free = np.ones(Iref.shape[1], bool) if fixed is None else ~np.asarray(fixed, bool)
    assert np.array_equal(I0[:, free], Iref[:, free]), 'step 0: sampled tokens differ'
    close(t0['Y'], torch.from_numpy(Y.reshape(b, -1)), 3e-6, 'step-0 confidences')
    Smax, tmax = np.zeros(b), np.zeros(b, int)
    active = np.ones(b, bool)
    checked = 0
    for rec in trace[1:]:
        t, k = rec['t'], rec['k']
        mask1 = rec['mask1'].cpu().numpy().astype(bool)
        Yb, Ib = rec['Y_before'].cpu().numpy(), rec['I_before'].cpu().numpy()
        tokn, Yn, _ = S.token_race(rec['logits'], rec['E_tok'])
        assert np.array_equal(tokn.reshape(b, Bm, -1), rec['Inew'].cpu().numpy()), f'step {t}: sampled tokens differ'
        close(rec['Ynew'], torch.from_numpy(Yn.reshape(b, Bm, -1)), 3e-6, f'step {t} confidences')
        assert np.array_equal(rec['active_before'].cpu().numpy().astype(bool), active)
        for i in range(b):
            for j in range(Bm):
                ref = S.keep_race(Yb[i], rec['E_keep'][i, j], fixed, k)
                assert np.array_equal(mask1[i, j], ref), f'step {t} video {i} cand {j}: keep mask differs'
            if not active[i]:
                assert np.array_equal(rec['I_tok'][i].cpu().numpy(), Ib[i])  # a stopped video is frozen
                continue
            Yr, Ir, Sr, jm = S.update(Yb[i], Ib[i], mask1[i], rec['Ynew'][i], rec['Inew'][i], rec['z_rel'][i * Bm:(i + 1) * Bm],
                                      rec['z_vid'][i * Bm:(i + 1) * Bm])
            got_S = rec['S'][i].cpu().numpy()
            assert np.allclose(got_S, Sr, rtol=1e-5)
            srt = np.sort(Sr)
            near_tie = Bm > 1 and srt[-1] - srt[-2] < 1e-6
            if not near_tie:
                assert int(rec['jmax'][i]) == jm, f'step {t} video {i}: best candidate {int(rec["jmax"][i])} vs {jm}'
                assert np.array_equal(rec['I_tok'][i].cpu().numpy(), Ir)
                assert np.allclose(rec['Y'][i].cpu().numpy(), Yr, rtol=0, atol=0)
            if dynamic:
                Smax[i], tmax[i], took, stop = S.dynamic_stop(float(Sr[jm]), t, Smax[i], tmax[i])
                if took:
                    assert np.array_equal(rec['Imax'][i].cpu().numpy(), rec['I_tok'][i].cpu().numpy())
                if stop:
                    active[i] = False
                assert bool(rec['active'][i]) == active[i], f'step {t} video {i}: dynamic stop differs'
            else:
                assert np.array_equal(rec['Imax'][i].cpu().numpy(), rec['I_tok'][i].cpu().numpy())
            checked += 1
    return checked


@pytest.mark.parametrize('Bm,dynamic,steps', [(1, False, 6), (2, True, 14)])
def test_mask_predict_trajectory_with_injected_randomness(golden, Bm, dynamic, steps):
    """SURVEY 8c (vi) on the HIP path: every sampled token, keep mask, candidate score / choice and dynamic stop of a
    batched mask-predict run equals what oracle/sampling.py decides from the same race variates and the same logits."""
    from oracle.synth import synth_tokens
    g = golden('bert_tiny')
    m = load_synth(tiny_bert(), g, 17).eval()
    text = synth_tokens('text', (3, 16), 49408, 17, low=1).to(DEV)
    mp = dict(golden('mask_predict').meta['mp_config'], B=Bm)
    gen = torch.Generator().manual_seed(5)

    def race(name, shape):
        return torch.empty(shape).exponential_(generator=gen).to(DEV)

    trace = []
    images, _, seq = m.generate_images(text, mask_predict_steps=steps, mp_config=mp, dynamic=dynamic, _race=race, _trace=trace)
    assert images.shape == (3, 2, 3, 64, 64) and seq.shape == (6, 16) and int(seq.max()) < 256
    n = _replay_trace(trace, None, 3, Bm, dynamic)
    print(f'mask-predict Bm={Bm} dynamic={dynamic}: {len(trace)} steps, {n} (video, step) updates verified')
    # a video's trajectory does not depend on its batch mates: video 1 alone with its own variates
    gen2 = torch.Generator().manual_seed(5)
    full = {}

    def race_rec(name, shape):
        full[name] = torch.empty(shape).exponential_(generator=gen2).to(DEV)
        return full[name]

    _, _, seq_a = m.generate_images(text, mask_predict_steps=steps, mp_config=mp, dynamic=dynamic, _race=race_rec)
    TS = m.target_seq_len

    def race_one(name, shape):
        v = full[name]
        if name.startswith('keep'):
            return v[1:2].contiguous()
        per = v.shape[0] // 3
        return v[per:2 * per].contiguous()

    _, _, seq_b = m.generate_images(text[1:2], mask_predict_steps=steps, mp_config=mp, dynamic=dynamic, _race=race_one)
    assert torch.equal(seq_a.view(3, TS)[1], seq_b.view(TS)), 'a video sampled alone differs from the same video in a batch'


def test_mask_predict_preserve_and_long_modes(golden):
    """dalle_bert.py:542-583: the 'long' (overlap frames carried over) and 'interp' (even frames given) modes."""
    g = golden('bert_tiny')
    m = load_synth(tiny_bert(), g, 17).eval()
    from oracle.synth import synth_tokens
    text = synth_tokens('text', (2, 16), 49408, 17, low=1).to(DEV)
    mp = dict(golden('mask_predict').meta['mp_config'], B=2)
    torch.manual_seed(0)
    prev = torch.randint(0, 256, (4, 16), device=DEV)  # [(b t), n]: the previous clip's tokens
    trace = []
    _, _, seq = m.generate_images(text, mask_predict_steps=5, mp_config=mp, dynamic=False, preserve=prev, t_overlap=1,
                                  long_mode='long', _trace=trace)
    seq = seq.view(2, 2, 16)
    assert torch.equal(seq[:, 0], prev.view(2, 2, 16)[:, 1])  # the last frame of the previous clip leads the new one
    fixed = np.zeros(32, bool)
    fixed[:16] = True
    _replay_trace(trace, fixed, 2, 2, False)
    assert all(r['k'] == 16 - s for r, s in zip(trace[1:], [14, 12, 11, 10]))  # N = 16: the schedule runs on the free part
    trace = []
    given = torch.randint(0, 256, (2, 32), device=DEV)
    _, _, seq = m.generate_images(text, mask_predict_steps=4, mp_config=mp, dynamic=True, preserve=given, long_mode='interp',
                                  _trace=trace)
    assert torch.equal(seq.view(2, 2, 16)[:, 0], given.view(2, 2, 16)[:, 0])
    _replay_trace(trace, fixed, 2, 2, True)
    with pytest.raises(RuntimeError):
        m.generate_images(text, mask_predict_steps=1, mp_config=mp)


def test_artv_sampler_distribution_and_cache_agreement(golden):
    """A2: the cached sampler's probability vector equals the reference's expression (top_k over all 51,584-style logits,
    softmax / temperature, dalle_artv.py:274-276); tokens drawn from injected race variates equal the oracle's; the
    KV-cache decoder and the full recomputation choose the same tokens from the same variates."""
    from mmvid_amd.dalle_artv import DALLE, top_k
    from oracle import sampling as S
    g = golden('artv_tiny')
    m = DALLE(dim=768, vae=tiny_vae(), cvae=None, num_text_tokens=49408, text_seq_len=16,
              which_transformer='openai_clip_visual', num_visuals=1, num_targets=2, transformer_layers=2)
    load_synth(m, g, 19).eval()
    text, tt = g['text'].to(DEV), g['target_tok'].to(DEV)
    from oracle import vqgan
    sd = synth_model_sd(g, 19)
    vt = vqgan.get_codebook_indices(sd, g['visual'].reshape(-1, 3, 64, 64), 64, 'vae.model.').view(2, -1).to(DEV)
    with torch.no_grad():
        full = m(text, visual=vt, target=tt[:, :5])[:, -1]  # [B, total_tokens], -max outside the image block
        c0, c1 = m._allowed_range(m.control_seq_len)
        for thres, temp in ((0.5, 1.0), (0.9, 0.7), (0.999, 1.0)):
            ref = F.softmax(top_k(full, thres=thres) / temp, dim=-1)  # the reference's lines, full width
            got = m.sampling_probs(full[:, c0:c1].contiguous(), thres, temp)
            assert ref[:, :c0].sum() == 0 and ref[:, c1:].sum() == 0
            close(got, ref[:, c0:c1], 1e-6, f'sampling distribution thres={thres} temp={temp}')
        E = torch.empty(2, c1 - c0).exponential_().to(DEV)
        tok = m._draw(full[:, c0:c1].contiguous(), 0.5, 0.8, lambda n, s: E, 'x')
        otok, _, _ = S.token_race(full[:, c0:c1], E, logit_div=0.8)
        assert np.array_equal(tok.view(-1).cpu().numpy(), otok)
    gen = torch.Generator().manual_seed(2)
    store = {}

    def race(name, shape):
        if name not in store:
            store[name] = torch.empty(shape).exponential_(generator=gen).to(DEV)
        return store[name]

    a = m.generate_images(text, visual=vt, _race=race)[0]
    b = m.generate_images(text, visual=vt, _race=race, use_cache=False)[0]
    assert a.shape == b.shape == (2, 2, 3, 64, 64)
    # informational: the two differ in bf16 summation order, and one flipped token changes the rest of that video (the
    # teacher-forced comparison of their logits is test_artv_kv_cache_decode_matches_full_recompute)
    print('cached vs recomputed sampling: identical videos', (a == b).flatten(1).all(1).float().mean().item())


def test_gemv_rows_kernel_vs_torch():
    """The decode-time linear layer (csrc/decode.hip): LN + bf16 weights streamed once + bias + QuickGELU + residual."""
    from mmvid_amd import ops
    torch.manual_seed(0)
    for NB, K, N, act, ln, res in ((1, 768, 2304, 0, True, False), (4, 768, 3072, 1, True, False), (3, 3072, 768, 0, False, True),
                                   (8, 768, 1024, 0, True, False), (2, 768, 770, 0, False, True), (8, 3072, 768, 0, False, True),
                                   (5, 3072, 768, 1, False, False), (19, 768, 1024, 0, True, True)):  # 19 rows: three launches
        x = torch.randn(NB, K, device=DEV)
        W = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
        b = torch.randn(N, device=DEV) * 0.1
        lw, lb = 1 + 0.1 * torch.randn(K, device=DEV), 0.1 * torch.randn(K, device=DEV)
        r = torch.randn(NB, N, device=DEV) if res else None
        y = ops.gemv_rows(x, W, b, ln=(lw, lb, 1e-5) if ln else None, act=act, residual=r, round_in=True)
        h = F.layer_norm(x, (K, ), lw, lb, 1e-5) if ln else x
        ref = h.bfloat16().float() @ W.float().t() + b
        if act:
            ref = ref * torch.sigmoid(1.702 * ref)
        if res:
            ref = ref + r
        close(y, ref, 2e-5, f'gemv NB={NB} K={K} N={N}')


def test_positional_table_and_weighted_loss_kernels(golden):
    """functional.PosTable (one launch each way) against the torch construction it replaces -- cat of special / text /
    per-visual axial (+ zero [SEP] rows) / target axial tables -- values bit-equal (same summation order), parameter
    gradients equal up to fp32 summation order; functional.weighted_loss against the arithmetic of train.py:320."""
    from mmvid_amd.functional import weighted_loss
    for nv, sep in ((0, False), (1, False), (2, True)):
        torch.manual_seed(nv)
        m = tiny_bert(nv, nv > 0, insert_sep=sep).to(DEV) if sep else tiny_bert(nv, nv > 0).to(DEV)
        names = [n for n, _ in m.named_parameters() if 'pos_emb' in n]
        assert names
        table = m._pos_table()
        sp = m.special_pos_emb.weight
        parts = [sp[0:1], m.text_pos_emb.weight[:m.text_seq_len]]
        if m.num_visuals > 0:
            parts.append(m.visual_pos_emb.table(insert_sep=bool(m.insert_sep)))
        parts += [sp[1:3], m.target_pos_emb.table()]
        ref = torch.cat(parts, 0)
        assert table.shape == ref.shape == (m.total_seq_len, 768) and torch.equal(table, ref)
        g = torch.randn_like(ref)
        want = torch.autograd.grad(ref, [p for n, p in m.named_parameters() if n in names], g, allow_unused=True)
        for p in m.parameters():
            p.grad = None
        table.backward(g)
        for n, w in zip(names, want):
            got = dict(m.named_parameters())[n].grad
            if w is None:
                assert got is None or float(got.abs().max()) == 0.0, n
            else:
                close(got, w, 1e-6, f'pos-table gradient of {n} (visuals {nv}, sep {sep})')
    a, b, c = [torch.tensor(v, device=DEV, requires_grad=True) for v in (1.25, -0.5, 3.0)]
    tot = weighted_loss((a, b, c), (7.0, 0.5, 0.25))
    assert float(tot) == 7.0 * 1.25 + 0.5 * -0.5 + 0.25 * 3.0
    (tot * 2.0).backward()
    assert (float(a.grad), float(b.grad), float(c.grad)) == (14.0, 1.0, 0.5)


# ------------------------------------------------------------------------------------------- heads, ids
def test_bert_build_ids_matches_torch(golden):
    from mmvid_amd import ops
    torch.manual_seed(0)
    B, Tt, Nv, TS = 4, 16, 16, 32
    text = torch.randint(1, 49408, (B, Tt))
    text[0, 9:] = 0
    text[2, 3:] = 0
    vis = torch.randint(0, 256, (B, Nv))
    tgt, warp = torch.randint(0, 256, (B, TS)), torch.randint(0, 256, (B, TS))
    mask1 = torch.rand(B, TS) < 0.4
    pad_base, MASK = 49408, 256
    for use_vis in (True, False):
        for rel, vid in ((True, True), (False, True), (True, False), (False, False)):
            ids, sel, tfull, cnt = ops.bert_build_ids(text.to(DEV), vis.to(DEV) if use_vis else None, Nv, tgt.to(DEV),
                                                      warp.to(DEV) if vid else None, mask1.to(DEV), pad_base, MASK, rel, vid)
            tx = torch.where(text == 0, torch.arange(Tt) + pad_base, text)
            vv = vis if use_vis else torch.full((B, Nv), MASK)
            ctrl = torch.cat([torch.zeros(B, 1, dtype=torch.long), tx, vv, torch.tensor([[1, 2]]).expand(B, 2)], 1)
            seqs = [torch.cat([ctrl, torch.where(mask1, tgt, MASK)], 1)]
            if rel:
                seqs.append(torch.cat([torch.cat([ctrl[B // 2:], ctrl[:B // 2]]), torch.where(mask1, tgt, MASK)], 1))
            if vid:
                seqs.append(torch.cat([ctrl, torch.where(mask1, warp, MASK)], 1))
            assert torch.equal(ids.cpu(), torch.cat(seqs, 0))
            L = ctrl.shape[1] + TS
            sel_ref = torch.cat([torch.zeros(B, ctrl.shape[1], dtype=torch.bool), ~mask1], 1)
            assert torch.equal(sel.cpu().bool().view(B, L), sel_ref) and cnt.item() == sel_ref.sum().item()
            assert torch.equal(tfull.cpu().view(B, L)[:, ctrl.shape[1]:], tgt)


def test_bert_heads_forward_backward_vs_torch():
    """functional.BertHeads (LN + GEMM + CE, LN + dot + BCE, one gradient tensor) against stock torch ops on the GPU."""
    from mmvid_amd.functional import BertHeads
    from mmvid_amd import ops
    torch.manual_seed(0)
    B, L, E, V, csl = 4, 51, 768, 256, 19
    TS = L - csl
    y = torch.randn(3 * B, L, E, device=DEV, requires_grad=True)
    P = lambda *s: torch.nn.Parameter(torch.randn(*s, device=DEV) * 0.05)
    lnw, lnb, W, bb = torch.nn.Parameter(1 + 0.1 * torch.randn(E, device=DEV)), P(E), P(V, E), P(V)
    rl = [torch.nn.Parameter(1 + 0.1 * torch.randn(E, device=DEV)), P(E), P(1, E), P(1)]
    vd = [torch.nn.Parameter(1 + 0.1 * torch.randn(E, device=DEV)), P(E), P(1, E), P(1)]
    mask1 = torch.rand(B, TS, device=DEV) < 0.3
    target = torch.randint(0, V, (B, TS), device=DEV)
    nfm = torch.tensor([1., 0., 1., 1.], device=DEV)
    sel = torch.cat([torch.zeros(B, csl, dtype=torch.bool, device=DEV), ~mask1], 1).reshape(-1).to(torch.uint8)
    tfull = torch.cat([torch.zeros(B, csl, dtype=torch.long, device=DEV), target], 1).reshape(-1).contiguous()
    cnt = sel.sum().float().reshape(1)
    ar = torch.arange(B, device=DEV)
    labels = torch.cat([torch.ones(B, device=DEV), torch.zeros(B, device=DEV)])
    rel_rows = torch.cat([ar * L, (B + ar) * L]).contiguous()
    vid_rows = torch.cat([ar * L + csl - 1, (2 * B + ar) * L + csl - 1]).contiguous()
    for by_nfm in (True, False):
        for p in [y, lnw, lnb, W, bb] + rl + vd:
            p.grad = None
        lm, lr, lv, logits = BertHeads.apply(y, tfull, sel, cnt, nfm, labels, rel_rows, vid_rows, by_nfm, B,
                                             ops.cast_bf16(W.detach()), lnw, lnb, W, bb, *rl, *vd)
        (7 * lm + 0.5 * lr + 0.5 * lv).backward()
        got = [lm.item(), lr.item(), lv.item()] + [p.grad.clone() for p in [y, lnw, lnb, W, bb] + rl + vd]
        for p in [y, lnw, lnb, W, bb] + rl + vd:
            p.grad = None
        out = y[:B]
        lg = F.linear(F.layer_norm(out[:, csl:], (E, ), lnw, lnb), W, bb)
        rm = F.cross_entropy(lg[~mask1], target[~mask1])
        head = lambda h, x: F.linear(F.layer_norm(x, (E, ), h[0], h[1]), h[2], h[3])
        lp, ln = head(rl, out[:, 0]).squeeze(), head(rl, y[B:2 * B, 0]).squeeze()
        vp, vn = head(vd, out[:, csl - 1]), head(vd, y[2 * B:, csl - 1])
        one, zero = torch.ones(B, device=DEV), torch.zeros(B, device=DEV)
        if by_nfm:
            den = max(1., nfm.sum().item())
            rr = (F.binary_cross_entropy_with_logits(lp, one, reduction='none') * nfm +
                  F.binary_cross_entropy_with_logits(ln, zero, reduction='none') * nfm).sum() / den
            rv = F.binary_cross_entropy_with_logits(vp, one[:, None], reduction='none').sum() / den + \
                F.binary_cross_entropy_with_logits(vn, zero[:, None], reduction='none').sum() / den
        else:
            rr = F.binary_cross_entropy_with_logits(lp, one) + F.binary_cross_entropy_with_logits(ln, zero)
            rv = F.binary_cross_entropy_with_logits(vp, one[:, None]) + F.binary_cross_entropy_with_logits(vn, zero[:, None])
        (7 * rm + 0.5 * rr + 0.5 * rv).backward()
        ref = [rm.item(), rr.item(), rv.item()] + [p.grad.clone() for p in [y, lnw, lnb, W, bb] + rl + vd]
        assert abs(got[0] - ref[0]) < 2e-2 * abs(ref[0]) and abs(got[1] - ref[1]) < 1e-5 and abs(got[2] - ref[2]) < 1e-5
        close(logits.view(B, L, V)[:, csl:], lg, 2e-2, 'logits')
        names = ['dy', 'dlnw', 'dlnb', 'dW', 'db', 'rel dlnw', 'rel dlnb', 'rel dw', 'rel db', 'vid dlnw', 'vid dlnb', 'vid dw', 'vid db']
        for nme, a, b in zip(names, got[3:], ref[3:]):
            close(a, b, 3e-2 if nme in ('dy', 'dlnw', 'dlnb', 'dW', 'db') else 2e-5, nme)
        # the REL / VID rows of dy alone (fp32 kernels): tight
        rows = torch.cat([rel_rows[B:], vid_rows[B:]])
        close(got[3].view(-1, E)[rows], ref[3].view(-1, E)[rows], 2e-5, 'dy rows of the negative passes')


# ------------------------------------------------------------------------------------------- device front-end
def _read_warp_params(raw, B):
    a = raw.cpu().numpy().view(np.int32).reshape(B, -1)
    f = raw.cpu().numpy().view(np.float32).reshape(B, -1)
    out = []
    for b in range(B):
        out.append(dict(mode=int(a[b, 0]), j1=int(a[b, 1]), src_b=int(a[b, 2]), src_t=int(a[b, 3]), chan=int(a[b, 4]),
                        shift=float(f[b, 5]), theta=torch.tensor(f[b, 6:12].reshape(2, 3).copy()), perm=a[b, 12:44].tolist()))
    return out


def test_frontend_vid_warp_matches_oracle():
    """The VID negative on the device (dalle_bert.py:204-238): parameters are drawn on the GPU; the resulting frames must
    equal oracle/frontend.py::apply_warp (torch affine_grid / grid_sample on the CPU) for those parameters, and the
    strategy / frame choices must follow the requested distribution."""
    from mmvid_amd.frontend import Frontend
    from oracle.frontend import apply_warp
    fe = Frontend(seed=1234)
    torch.manual_seed(0)
    B, T = 64, 8
    x = torch.rand(B, T, 3, 32, 32)
    modes = []
    for rep in range(6):
        out = fe.vid_warp(x.to(DEV), [0.25, 0.25, 0.25, 0.25])
        params = _read_warp_params(fe._warp_scratch[:B * 176], B)
        for p in params:
            p['perm'] = p['perm'][:T]
            assert 0 <= p['j1'] < T and 0 <= p['src_t'] < T and 0 <= p['src_b'] < B
            if p['mode'] == 1:
                assert sorted(p['perm']) == list(range(T)) and p['perm'] != list(range(T))
            if p['mode'] == 2:
                assert -0.5 <= p['shift'] < 0.5 and 0 <= p['chan'] <= 3
            if p['mode'] == 3:
                sc = math.hypot(float(p['theta'][0, 0]), float(p['theta'][1, 0]))
                assert 0.9 <= sc <= 1.1 + 1e-6 and abs(float(p['theta'][0, 2])) <= 0.1 and abs(float(p['theta'][1, 2])) <= 0.1
                assert abs(math.atan2(float(p['theta'][1, 0]), float(p['theta'][0, 0]))) <= math.pi / 6 + 1e-6
        for b, p in enumerate(params):
            if p['mode'] == 0:
                assert p['src_b'] != b
        ref = apply_warp(x, params)
        close(out, ref, 2e-5, f'warped frames (draw {rep})')
        modes += [p['mode'] for p in params]
        fe.advance(DEV)
    freq = np.bincount(modes, minlength=4) / len(modes)
    print('warp strategy frequencies', freq)
    assert np.abs(freq - 0.25).max() < 0.08
    # a different step -> different draws; the same step -> the same draws
    a = fe.vid_warp(x.to(DEV), [0.25] * 4).clone()
    assert torch.equal(a, fe.vid_warp(x.to(DEV), [0.25] * 4))
    fe.advance(DEV)
    assert not torch.equal(a, fe.vid_warp(x.to(DEV), [0.25] * 4))


def test_vid_negative_tokens_without_reencoding():
    """BERT.forward encodes the B*T target frames plus ONE new frame per sample and assembles the VID negative's tokens
    (frontend.vid_warp_tokens) instead of tokenising the whole warped video again.  Must be bit-identical to the
    reference's order of operations: warp the pixels (dalle_bert.py:1094), then tokenise all of them (1095)."""
    from mmvid_amd.frontend import Frontend
    torch.manual_seed(0)
    vae = tiny_vae().to(DEV)
    with torch.no_grad():
        vae.model.quantize.embedding.weight.normal_(0, 0.5)
    fe = Frontend(seed=5)
    B, T = 8, 4
    x = torch.rand(B, T, 3, 64, 64, device=DEV)
    seen = set()
    for rep in range(6):
        full = fe.vid_warp(x, [0.25] * 4)  # draws the parameters of this step
        ref = vae.get_codebook_indices(full.view(B * T, 3, 64, 64)).view(B, -1)
        both = torch.empty(B * T + B, 3, 64, 64, device=DEV)
        both[:B * T].copy_(x.view(B * T, 3, 64, 64))
        fe.vid_warp_new_frames(x, [0.25] * 4, both[B * T:])  # same (seed, step): the same parameters
        toks = vae.get_codebook_indices(both)
        got = fe.vid_warp_tokens(toks[:B * T].reshape(B, -1).contiguous(), toks[B * T:], T)
        assert torch.equal(got, ref), f'draw {rep}: {(got != ref).sum().item()} tokens differ'
        seen |= {p['mode'] for p in _read_warp_params(fe._warp_scratch[:B * 176], B)}
        fe.advance(DEV)
    assert seen == {0, 1, 2, 3}


def test_frontend_msm_masks_distribution():
    """MSM masking strategies on the device (dalle_bert.py:992-1029) against oracle/frontend.py: strategy frequencies,
    Bernoulli keep rate, erased-box area / shape statistics, frame preservation."""
    from mmvid_amd.frontend import Frontend
    from oracle.frontend import msm_masks
    fe = Frontend(seed=7)
    T, f, B = 8, 8, 256
    prob, bern = [0.7, 0.1, 0.1, 0.1], [0.2, 0.5]
    M, NF, ST = [], [], []
    for _ in range(12):
        m, nfm, st = fe.msm_masks(B, T, f, DEV, prob, bern, 0.0, want_strategy=True)
        M.append(m.cpu().bool()), NF.append(nfm.cpu()), ST.append(st.cpu().numpy())
        fe.advance(DEV)
    M, NF, ST = torch.cat(M), torch.cat(NF), np.concatenate(ST)
    rng = np.random.RandomState(0)
    Mo, NFo, STo = msm_masks(rng, len(ST), T, f, prob, bern)
    freq, freqo = np.bincount(ST, minlength=5)[1:] / len(ST), np.bincount(STo, minlength=5)[1:] / len(STo)
    print('strategy frequencies device', freq, 'oracle', freqo)
    assert np.abs(freq - np.array(prob)).max() < 0.04
    assert ((NF == 0).numpy() == (ST == 2)).all() and (~M[ST == 2]).all()

    def box_stats(Mx, STx, which):
        sel = Mx[STx == which].view(-1, T, f, f)
        assert (sel == sel[:, :1]).all()  # one box for all frames
        hid = ~sel[:, 0] if which == 3 else sel[:, 0]
        area = hid.flatten(1).float().sum(1)
        rows, cols = hid.any(2).float().sum(1), hid.any(1).float().sum(1)
        assert (area == rows * cols).all()  # rectangles
        return area.mean().item() / (f * f), (rows / cols.clamp(min=1))[area > 0].log().mean().item(), (area == 0).float().mean().item()

    for which in (3, 4):
        a, r, e = box_stats(M, ST, which)
        ao, ro, eo = box_stats(Mo, STo, which)
        print(f'strategy {which}: area fraction {a:.3f} (oracle {ao:.3f}), mean log aspect {r:.3f} ({ro:.3f}), no-box rate {e:.3f} ({eo:.3f})')
        assert abs(a - ao) < 0.05 and abs(r - ro) < 0.15 and abs(e - eo) < 0.08
    k1, k1o = M[ST == 1].float().mean().item(), Mo[STo == 1].float().mean().item()
    assert abs(k1 - 0.35) < 0.02 and abs(k1o - 0.35) < 0.02
    # frame preservation: with pc_prob = 1 between 1 and T/2 whole frames of every sample are visible
    m, _, st = fe.msm_masks(512, T, f, DEV, [0, 1, 0, 0], bern, 1.0, want_strategy=True)
    kept = m.cpu().bool().view(512, T, f * f).all(2).sum(1)
    assert kept.min() >= 1 and kept.max() <= T // 2 and set(kept.tolist()) == {1, 2, 3, 4}


def test_frontend_token_erasing():
    from mmvid_amd.frontend import Frontend, face_choices
    fe = Frontend(seed=3)
    B, Tv, f = 6, 2, 8
    tok = torch.randint(0, 1024, (B, Tv * f * f), device=DEV)
    for mode, fm in (('face_8x8', 'eyes_nose'), ('face_8x8', 'mouth'), ('face2_8x8', None), ('mask_8x8', 'fixed'), ('shape_4x4', None)):
        ch, f0 = face_choices(mode, fm)
        out = fe.erase_choice(tok.clone(), Tv, f, 1024, ch, f0).view(B, Tv, f, f).cpu()
        src = tok.view(B, Tv, f, f).cpu()
        (_, md, (r0, r1, c0, c1)), = ch
        inside = torch.zeros(f, f, dtype=torch.bool)
        inside[r0:r1, c0:c1] = True
        for t in range(Tv):
            if f0 and t == 0:
                assert torch.equal(out[:, t], src[:, t])
            elif md == 1:
                assert torch.equal(out[:, t][:, inside], src[:, t][:, inside]) and (out[:, t][:, ~inside] == 1024).all()
            else:
                assert (out[:, t][:, inside] == 1024).all() and torch.equal(out[:, t][:, ~inside], src[:, t][:, ~inside])
    # the random three-way choice of mask_8x8 (0.5 / 0.25 / 0.25), one draw per call
    seen = []
    for _ in range(200):
        out = fe.erase_choice(tok.clone(), Tv, f, 1024, *face_choices('mask_8x8', None)).view(B, Tv, f, f)
        seen.append(int((out == 1024).all(0).all(0).sum().item()))
        fe.advance(DEV)
    fr = [seen.count(0) / 200, seen.count(48) / 200, seen.count(28) / 200]  # untouched / 4x4 kept / 6x6 kept
    print('mask_8x8 choice frequencies', fr)
    assert abs(fr[0] - 0.5) < 0.12 and abs(fr[1] - 0.25) < 0.1 and abs(fr[2] - 0.25) < 0.1
    out = fe.random_erase(tok.clone(), Tv, f, 1024, 1.0, (0.55, 0.85), (0.5, 2.0), True).view(B, Tv, f, f)
    assert (out[:, :, f // 2:] == 1024).all() and torch.equal(out[:, :, :f // 2].cpu(), tok.view(B, Tv, f, f)[:, :, :f // 2].cpu())
    out = fe.random_erase(tok.clone(), Tv, f, 1024, 1.0, (0.2, 0.5), (0.5, 2.0), False).view(B, Tv, f, f)
    hid = out == 1024
    assert (hid[:, 0] == hid[:, 1]).all() and hid.any()


def test_bert_forward_with_device_frontend_trains(golden):
    """No injected randomness: masks, warp and erasing come from the device front-end; the step is deterministic for a
    (seed, step) pair, differs between steps, and the loss goes down under FlatTrainer with the warm-up schedule."""
    from mmvid_amd.engine import FlatTrainer, WarmupLR, backward_order
    g = golden('bert_tiny_visual')
    m = load_synth(tiny_bert(1, True), g, 17).train()
    text, frames, visual = g['text'].to(DEV), g['frames'].to(DEV), g['visual'].to(DEV)
    kw = dict(visual=visual, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True, vc_mode='shape_4x4',
              erase_visual=True, msm_strategy_prob=[0.7, 0.1, 0.1, 0.1], msm_bernoulli_prob=[0.2, 0.2], pc_prob=0.2)
    m.frontend.seed, m.frontend.step = 99, None
    with torch.no_grad():
        a = [x.item() for x in m(text, **kw)]
        b = [x.item() for x in m(text, **kw)]
        m.frontend.step = None
        c = [x.item() for x in m(text, **kw)]
    assert all(math.isfinite(v) for v in a + b)
    assert max(abs(x - y) for x, y in zip(a, c)) < 1e-4, (a, c)  # same (seed, step): same masks, warp, erasing
    assert max(abs(x - y) for x, y in zip(a, b)) > 1e-3, (a, b)  # next step: new draws
    tr = FlatTrainer(m, lr=1e-3, max_grad_norm=1.0, order=backward_order, lr_schedule=WarmupLR(1e-6, 1e-3, 8, every=1))
    losses = []
    for i in range(12):
        tr.zero_grad()
        lm, lr, lv = m(text, **kw)
        (7 * lm + 0.5 * lr + 0.5 * lv).backward()
        tr.step()
        losses.append(lm.item())
        want = WarmupLR(1e-6, 1e-3, 8, every=1).lr_at(i)
        assert abs(tr._lr_dev.item() - want) <= 1e-6 * want + 1e-12, (i, tr._lr_dev.item(), want)
    print('msm losses with the device front-end', [round(v, 3) for v in losses])
    assert min(losses[-4:]) < max(losses[:3])


# ------------------------------------------------------------------------------------------- engine / boundary
def test_flat_trainer_state_dict_roundtrip_and_bindings(golden):
    from mmvid_amd.engine import FlatTrainer, backward_order
    g = golden('bert_tiny')
    m = load_synth(tiny_bert(), g, 17).train()
    text, frames = g['text'].to(DEV), g['frames'].to(DEV)

    def step(tr, model):
        tr.zero_grad()
        lm, lr, lv = _with_tokens(model, g, lambda: model(text, target=frames, return_loss=True, rel=True, vid=True,
                                                          _mask1=g['mask1'], _target_warp=g['warped_frames']))
        (7 * lm + 0.5 * lr + 0.5 * lv).backward()
        tr.step()

    tr = FlatTrainer(m, lr=1e-3, order=backward_order)
    step(tr, m), step(tr, m)
    sd_opt, sd_model = tr.state_dict(), {k: v.clone() for k, v in m.state_dict().items()}
    assert set(sd_opt) >= {'state', 'param_groups'} and sd_opt['state'][0]['exp_avg'].shape == tr.params[0].shape
    step(tr, m)
    after3 = tr.P.clone()
    # resume from the checkpoint in a fresh model / trainer: the third step must reproduce
    m2 = load_synth(tiny_bert(), g, 17).train()
    tr2 = FlatTrainer(m2, lr=1e-3, order=backward_order)
    m2.load_state_dict(sd_model)
    tr2.load_state_dict(sd_opt)
    tr2.refresh_shadows()
    assert tr2.step_count == 2 and tr2._step_dev.item() == 2.0
    step(tr2, m2)
    close(tr2.P, after3, 2e-4, 'parameters after resume + 1 step vs uninterrupted run')
    # load_state_dict AFTER the trainer attached its shadows: the heads' bf16 view follows (advisor finding)
    lin = m2.to_logits[1]
    assert torch.equal(m2._w16(lin), lin.weight.detach().bfloat16())
    with torch.no_grad():
        lin.weight.mul_(1.5)
    assert torch.equal(m2._w16(lin), lin.weight.detach().bfloat16()) and m2._w16(lin).data_ptr() == tr2._shadow_view(lin.weight).data_ptr()
    # model.zero_grad(set_to_none=True) detaches .grad from G: the next step re-binds instead of training on zeros
    tr2.zero_grad()
    m2.zero_grad(set_to_none=True)
    before = tr2.P.clone()
    lm, lr, lv = _with_tokens(m2, g, lambda: m2(text, target=frames, return_loss=True, rel=True, vid=True, _mask1=g['mask1'],
                                                  _target_warp=g['warped_frames']))
    (7 * lm + 0.5 * lr + 0.5 * lv).backward()
    tr2.step()
    assert not torch.equal(before, tr2.P) and all(p.grad.data_ptr() == tr2.G.data_ptr() + 4 * o for p, o in zip(tr2.params, tr2.offsets))


def test_graphed_step_with_captured_gradient_exchange(golden):
    """The data-parallel step keeps the graph: with a process group (here a group of ONE rank over RCCL, exchange forced
    on) the bucketed all-reduces issued from the tower-backward callbacks are captured inside the step's hipGraph.
    Parameters after three steps must equal the plain single-process run (a sum over one rank is the identity)."""
    import copy
    import socket

    import torch.distributed as dist

    from mmvid_amd.engine import FlatTrainer, GraphedStep, backward_order
    g = golden('bert_tiny')
    base = load_synth(tiny_bert(), g, 17).train()
    text, frames = g['text'].to(DEV), g['frames'].to(DEV)
    mask1, warped = g['mask1'].to(DEV), g['warped_frames'].to(DEV)
    nfm = torch.ones(text.shape[0], device=DEV)

    def run(force):
        m = copy.deepcopy(base)
        m.transformer.backward_chunk_layers = 1
        tr = FlatTrainer(m, lr=1e-3, order=backward_order, force_exchange=force, bucket_mb=4)

        def fn(text, frames):  # capture-safe: device work only (the model's own encoder, injected mask / warp tensors)
            lm, lr, lv = m(text, target=frames, return_loss=True, rel=True, vid=True, _mask1=mask1, _target_warp=warped,
                           _not_fully_masked=nfm)
            return 7.0 * lm + 0.5 * lr + 0.5 * lv

        step = GraphedStep(tr, fn, dict(text=text, frames=frames), warmup=1)
        losses = [step().item() for _ in range(2)]
        return step, tr, losses

    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        step_f, tr_f, loss_f = run(True)
        assert tr_f.force_exchange
        print('captured exchange:', 'hipGraph' if step_f.graph is not None else f'eager fallback ({step_f.capture_error})')
        step_p, tr_p, loss_p = run(False)
        assert step_p.graph is not None
        assert tr_f.step_count == tr_p.step_count == 3
        for a, b in zip(loss_f, loss_p):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(a))
        close(tr_f.P, tr_p.P, 5e-3, 'parameters after 3 steps: forced exchange vs plain')
        assert step_f.graph is not None, f'the step with RCCL all-reduces could not be captured: {step_f.capture_error}'
    finally:
        dist.destroy_process_group()


def test_graphed_step_falls_back_to_eager_when_capture_fails(golden):
    """If the step cannot be captured (here: a host read inside it; on a multi-GPU node it could be a collective the runtime
    refuses to capture) GraphedStep reports why and keeps training with eager launches: same losses as a plain eager loop."""
    import copy

    from mmvid_amd.engine import FlatTrainer, GraphedStep, backward_order
    g = golden('bert_tiny')
    base = load_synth(tiny_bert(), g, 17).train()
    text, frames = g['text'].to(DEV), g['frames'].to(DEV)
    mask1, warped = g['mask1'].to(DEV), g['warped_frames'].to(DEV)
    nfm = torch.ones(text.shape[0], device=DEV)

    def run(host_read, graphed):
        m = copy.deepcopy(base)
        tr = FlatTrainer(m, lr=1e-3, order=backward_order)

        def fn(text, frames):
            lm, lr, lv = m(text, target=frames, return_loss=True, rel=True, vid=True, _mask1=mask1, _target_warp=warped,
                           _not_fully_masked=nfm)
            if host_read:
                float(lm)  # a device->host synchronisation: illegal during stream capture
            return 7.0 * lm + 0.5 * lr + 0.5 * lv

        if not graphed:
            out = []
            for _ in range(4):
                tr.zero_grad()
                loss = fn(text, frames)
                loss.backward()
                tr.step()
                out.append(loss.item())
            return None, out
        step = GraphedStep(tr, fn, dict(text=text, frames=frames), warmup=1)
        return step, [step().item() for _ in range(3)]

    step, got = run(True, True)
    assert step.graph is None and step.capture_error, 'a step with a host read must not have been captured'
    print('capture refused as expected:', step.capture_error)
    _, want = run(False, False)
    for a, b in zip(got, want[1:]):  # the graphed run spent its first step on warm-up
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (got, want)
    step2, again = run(False, True)  # and capturing still works afterwards in the same process
    assert step2.graph is not None
    for a, b in zip(again, want[1:]):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (again, want)


def test_unchanged_train_loop_under_torch_ddp(golden):
    """train.py:28-35 wraps the model in DistributedDataParallel(find_unused_parameters=True) and steps a torch optimiser.
    The kernels write parameter gradients straight into p.grad, so under DDP every Function also hands autograd a zero
    gradient per parameter: the reducer's hooks fire and the step completes.  Same parameters as the unwrapped model."""
    import copy
    import socket

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    g = golden('bert_tiny')
    base = load_synth(tiny_bert(), g, 17).train()
    text, frames = g['text'].to(DEV), g['frames'].to(DEV)
    mask1, warped = g['mask1'].to(DEV), g['warped_frames'].to(DEV)
    nfm = torch.ones(text.shape[0], device=DEV)

    def run(wrap):
        m = copy.deepcopy(base)
        model = DDP(m, device_ids=[0], find_unused_parameters=True) if wrap else m
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        losses = []
        for _ in range(3):
            lm, lr, lv = model(text, target=frames, return_loss=True, rel=True, vid=True, _mask1=mask1, _target_warp=warped,
                               _not_fully_masked=nfm)
            loss = 7.0 * lm + 0.5 * lr + 0.5 * lv
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opt.step()
            losses.append(loss.item())
        return losses, torch.cat([p.detach().flatten()[::97] for p in m.parameters() if p.requires_grad])

    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        lw, pw = run(True)
    finally:
        dist.destroy_process_group()
    lp, pp = run(False)
    print('DDP-wrapped losses', lw, 'plain', lp)
    for a, b in zip(lw, lp):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b))
    close(pw, pp, 5e-3, 'sampled parameters after 3 Adam steps: DDP-wrapped vs plain')


def test_artv_flat_trainer_keeps_head_shadow_current(golden):
    """Advisor finding: DALLE's 51,584-way head must train against a bf16 weight that follows the fused optimiser."""
    from mmvid_amd.dalle_artv import DALLE
    from mmvid_amd.engine import FlatTrainer, backward_order
    from oracle import vqgan
    g = golden('artv_tiny')
    m = DALLE(dim=768, vae=tiny_vae(), cvae=None, num_text_tokens=49408, text_seq_len=16,
              which_transformer='openai_clip_visual', num_visuals=1, num_targets=2, transformer_layers=2)
    load_synth(m, g, 19).train()
    sd = synth_model_sd(g, 19)
    vt = vqgan.get_codebook_indices(sd, g['visual'].reshape(-1, 3, 64, 64), 64, 'vae.model.').view(2, -1).to(DEV)
    text, tt = g['text'].to(DEV), g['target_tok'].to(DEV)
    tr = FlatTrainer(m, lr=1e-3, order=backward_order)
    losses = []
    for _ in range(4):
        tr.zero_grad()
        loss = m(text, visual=vt, target=tt, return_loss=True)[0]
        loss.backward()
        tr.step()
        losses.append(loss.item())
        assert torch.equal(m._w16(), m.to_logits[1].weight.detach().bfloat16())
    assert losses[-1] < losses[0]
    # erase_codebook_face for ART-V (dalle_artv.py:356-416): erased positions become per-position pad ids
    out = m.erase_codebook_face(vt.clone(), 'face2_8x8' if m.image_fmap_size == 8 else 'shape_4x4')
    assert ((out == -1) | (out == vt)).all() and (out == -1).any()
    assert torch.isfinite(m(text, visual=vt, target=tt, return_loss=True, vc_mode='shape_4x4')[0])


def test_tower_with_frozen_matrices_still_backpropagates():
    from mmvid_amd.clip_tower import OpenAICLIPTransformer
    torch.manual_seed(0)
    tw = OpenAICLIPTransformer(40, 'openai_clip_visual', layers=2).to(DEV).train()
    x = torch.randn(2, 40, 768, device=DEV, requires_grad=True)
    gy = torch.randn(2, 40, 768, device=DEV)
    tw(x).backward(gy)
    ref = x.grad.clone()
    for p in tw._matrix_params():
        p.requires_grad_(False)
        p.grad = None
    x.grad = None
    tw(x).backward(gy)
    assert torch.equal(x.grad, ref) and all(p.grad is None for p in tw._matrix_params())


# ------------------------------------------------------------------------------------------- full-size configurations
def _full_bert(num_visuals, layers=12, text_len=64):
    from mmvid_amd.dalle_bert import BERT
    from mmvid_amd.vae import VQGanVAE1024
    torch.manual_seed(0)
    vae = VQGanVAE1024(None, 128)
    vae.image_size = 128
    cvae = None
    if num_visuals:
        cvae = VQGanVAE1024(None, 128)
        cvae.image_size = 128
    m = BERT(dim=768, vae=vae, cvae=cvae, num_text_tokens=49408, text_seq_len=text_len, which_transformer='openai_clip_visual',
             num_visuals=num_visuals, num_targets=8, transformer_layers=layers).to(DEV).train()
    with torch.no_grad():
        for v in (vae, cvae):
            if v is not None:
                v.model.quantize.embedding.weight.normal_(0, 0.5)
    return m


def test_config2_full_size_graphed_step_matches_eager():
    """BASELINE config 2 (12 layers, L = 579, per-GPU batch 6, 96 VQGAN frames per step) under test, not only in bench.py:
    the captured step (GraphedStep) and the eagerly launched step agree on loss and on sampled gradients / parameters."""
    import copy

    from mmvid_amd.engine import FlatTrainer, GraphedStep, backward_order
    base = _full_bert(0)
    assert base.total_seq_len == 579
    B = 6
    gen = torch.Generator().manual_seed(1)
    text = torch.randint(1, 49408, (B, 64), generator=gen)
    text[0, 40:] = 0
    data = [(torch.rand(B, 8, 3, 128, 128, generator=gen), torch.rand(B, 512, generator=gen) < 0.3,
             torch.rand(B, 8, 3, 128, 128, generator=gen)) for _ in range(3)]
    nfm = torch.ones(B, device=DEV)

    def run(graph):
        m = copy.deepcopy(base)
        tr = FlatTrainer(m, lr=1e-4, max_grad_norm=1.0, order=backward_order)

        def fn(text, frames, mask1, warped):
            lm, lr, lv = m(text, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True, _mask1=mask1,
                           _not_fully_masked=nfm, _target_warp=warped)
            return 7.0 * lm + 0.5 * lr + 0.5 * lv

        inp = lambda i: dict(text=text.to(DEV), frames=data[i][0].to(DEV), mask1=data[i][1].to(DEV), warped=data[i][2].to(DEV))
        losses = []
        if graph:
            step = GraphedStep(tr, fn, inp(0), warmup=1)
            for i in (1, 2):
                losses.append(step(**inp(i)).item())
        else:
            for i in (0, 1, 2):
                tr.zero_grad()
                loss = fn(**inp(i))
                loss.backward()
                tr.step()
                losses.append(loss.item())
            losses = losses[1:]
        return losses, tr.P[::997].clone(), tr.G[::997].clone()

    le, pe, ge = run(False)
    lg, pg, gg = run(True)
    print('config 2 full size: eager losses', le, 'graphed', lg)
    for a, b in zip(le, lg):
        assert math.isfinite(a) and abs(a - b) <= 2e-3 * max(1.0, abs(a))
    close(gg, ge, 2e-2, 'sampled gradients of the last step: graphed vs eager')
    close(pg, pe, 1e-3, 'sampled parameters after 3 steps: graphed vs eager')


@pytest.mark.parametrize('layers', [4, 12])
def test_config4_full_size_text_and_mask(layers):
    """BASELINE config 4: text_and_mask, one visual control frame through the cvae, L = 643 with the restricted rows at
    129 / 130; a full step with the device front-end (vc_mode mask_8x8 as scripts/mmvoxceleb/text_and_mask/train.sh).
    layers = 12 is the model of the config, 4 the fast variant."""
    from mmvid_amd.engine import FlatTrainer, backward_order
    m = _full_bert(1, layers=layers)
    assert m.total_seq_len == 643 and (m.st1_tok_index, m.vid_tok_index) == (129, 130)
    assert m.transformer.mask_spec == ('rows', [(129, 129), (130, 130)])
    B = 4
    torch.manual_seed(2)
    text = torch.randint(1, 49408, (B, 64), device=DEV)
    frames, visual = torch.rand(B, 8, 3, 128, 128, device=DEV), torch.rand(B, 1, 3, 128, 128, device=DEV)
    tr = FlatTrainer(m, lr=1e-4, order=backward_order)
    losses = []
    for _ in range(3):
        tr.zero_grad()
        lm, lr, lv = m(text, visual=visual, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                       vc_mode='mask_8x8', msm_strategy_prob=[0.7, 0.1, 0.1, 0.1], msm_bernoulli_prob=[0.2, 0.2])
        loss = 7 * lm + 0.5 * lr + 0.5 * lv
        loss.backward()
        tr.step()
        losses.append([lm.item(), lr.item(), lv.item()])
    print('config 4 losses', losses)
    assert all(math.isfinite(v) for l in losses for v in l)
    assert m.visual_emb.weight.grad.abs().sum() > 0 and m.visual_pos_emb.module_list[0].weights_0.grad.abs().sum() > 0
    # the control embedding keeps the visual segment where the reference puts it
    with torch.no_grad():
        ce = m(text, visual=visual, return_loss=False)
    assert ce.shape == (B, 131, 768)


@pytest.mark.parametrize('layers', [4, 12])
def test_config5_full_size_artv_forward_and_cached_decode(layers):
    """BASELINE config 5: ART-V, 16 frames, L = 1152, 51,584 classes.  Training loss + backward at full length, then 64
    cached decode steps against full recomputation of the same prefixes.  layers = 12 is the model of the config."""
    from mmvid_amd.dalle_artv import DALLE
    from mmvid_amd.vae import VQGanVAE1024
    torch.manual_seed(0)
    vae = VQGanVAE1024(None, 128)
    vae.image_size = 128
    m = DALLE(dim=768, vae=vae, cvae=None, num_text_tokens=49408, text_seq_len=64, which_transformer='openai_clip_visual',
              num_visuals=1, num_targets=16, transformer_layers=layers).to(DEV)
    assert m.total_seq_len == 1152 and m.total_tokens == 51584
    B = 2
    text = torch.randint(1, 49408, (B, 64), device=DEV)
    vt = torch.randint(0, 1024, (B, 64), device=DEV)
    tt = torch.randint(0, 1024, (B, 1024), device=DEV)
    m.train()
    loss = m(text, visual=vt, target=tt, return_loss=True)[0]
    loss.backward()
    ref = math.log(49472) / 9 + math.log(1088) / 9 + 7 * math.log(1024) / 9
    print('config 5 loss', loss.item(), 'uniform-prediction value', ref)
    assert math.isfinite(loss.item()) and abs(loss.item() - ref) < 1.5
    gw = m.to_logits[1].weight.grad
    assert gw[:49472].abs().sum() > 0 and gw[49472:50560].abs().sum() > 0 and gw[50560:].abs().sum() > 0
    m.eval()
    with torch.no_grad():
        prompt = torch.cat(m._prompt_ids(text, vt), 1)
        cache = m.transformer.new_kv_cache(B, m.total_seq_len, DEV)
        h = m.transformer.prefill(m._embed_rows(prompt, 0), cache)[:, -1, :]
        sess = m.transformer.decode_session(cache, prompt.shape[1])
        c0, c1 = m._allowed_range(m.control_seq_len)
        worst = 0.0
        for k in range(65):
            inc = m._logits_rows(h.contiguous(), (c0, c1))
            if k in (0, 1, 17, 40, 64):
                full = m(text, visual=vt, target=tt[:, :k])[:, -1, c0:c1]
                worst = max(worst, relerr(inc.cpu(), full.cpu()))
            h = sess.step(m._embed_rows(tt[:, k:k + 1], prompt.shape[1] + k)[:, 0, :])
        print('config 5: cached vs recomputed logits over 64 decode steps, worst relative error', worst)
        assert worst < 2e-2
