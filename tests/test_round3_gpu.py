"""Round-3 parity pins on a real MI355X: the device front-end and the samplers against outputs of the REFERENCE's own
functions (tests/golden/frontend.npz, mask_predict_race.npz: tools/make_golden.py::case_frontend / case_mask_predict_race),
the remaining BERT options (stable, motion_color, width-512 tower), and the data-parallel row-wise exchange with a
simulated second rank."""
import math

import numpy as np
import pytest
import torch

from conftest import relerr
from test_host_logic import tiny_bert
from test_models_gpu import DEV, close, load_synth

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------- H6 / N2: front-end vs the reference
def _pack_warp_params(dec, T):
    """golden `warp_decisions` rows -> the kernel's WarpParams records (176 B: 5 int, shift, theta[6], perm[32])."""
    from oracle.frontend import affine_theta
    B = dec.shape[0]
    raw = np.zeros((B, 44), np.int32)
    f = raw.view(np.float32)
    for b, r in enumerate(dec.tolist()):
        mode = int(r[0])
        raw[b, 0], raw[b, 1], raw[b, 2], raw[b, 3], raw[b, 4] = mode, int(r[1]), int(r[2]) if mode == 0 else b, int(r[3]), int(r[4])
        f[b, 5] = np.float32(r[5])
        if mode == 3:
            f[b, 6:12] = affine_theta(*[float(np.float32(v)) for v in r[6:10]]).reshape(-1).numpy()
        raw[b, 12:44] = np.arange(32)
        if mode == 1:
            raw[b, 12:12 + T] = [int(v) for v in r[10:10 + T]]
    return torch.from_numpy(raw.view(np.uint8).reshape(-1).copy())


def test_device_vid_warp_equals_reference_warp(golden):
    """mmvid_vid_warp applied to the decisions the reference's `warp` drew (dalle_bert.py:204-238) gives the reference's
    warped clip: frame swap / shuffle / colour shift exactly, the affine resampling within fp32 round-off."""
    from mmvid_amd.frontend import Frontend
    g = golden('frontend')
    x = g['clip'].to(DEV)
    fe = Frontend(seed=0)
    seen = set()
    for dec, ref in zip(g['warp_decisions'], g['warp_out']):
        params = _pack_warp_params(dec, x.shape[1]).to(DEV)
        out = fe.vid_warp(x, [0.25] * 4, params=params).cpu()
        for b in range(x.shape[0]):
            mode = int(dec[b, 0])
            seen.add(mode)
            if mode == 3:
                assert relerr(out[b], ref[b]) <= 2e-5, (b, relerr(out[b], ref[b]))
            else:
                assert torch.equal(out[b], ref[b]), (b, mode)
        # the one-new-frame form used by BERT.forward: the new frame equals the reference's frame j1
        nf = torch.empty(x.shape[0], *x.shape[2:], device=DEV)
        import mmvid_amd._lib as L
        from mmvid_amd.ops import _p, _stream
        L.call('mmvid_vid_warp_new_frames', 0, None, _p(x), x.shape[0], x.shape[1], 3, x.shape[3], x.shape[4],
               (L.F * 4)(0.25, 0.25, 0.25, 0.25), _p(params), 0, _p(nf), _stream())
        for b in range(x.shape[0]):
            if int(dec[b, 0]) in (2, 3):
                assert relerr(nf[b].cpu(), ref[b, int(dec[b, 1])]) <= 2e-5
    assert seen == {0, 1, 2, 3}


def test_device_msm_kernel_on_reference_decisions(golden):
    """The MSM mask kernel fed the decisions of a reference BERT.forward run (strategy, Bernoulli field, RandomErasing box,
    preserved frames; dalle_bert.py:992-1029) reproduces the reference's mask1 and not_fully_masked."""
    from mmvid_amd.frontend import Frontend
    g = golden('frontend')
    T, f = g.meta['T'], g.meta['fmap']
    strat, box, keep = g['msm_strategy'], g['msm_box'], g['msm_keep_frames']
    B = strat.shape[0]
    dec = torch.zeros(B, 72, dtype=torch.int32)
    dec[:, 0] = strat.int()
    dec[:, 1] = (strat >= 3).int()
    dec[:, 2:6] = box.int()
    dec[:, 8:8 + T] = (keep > 0).int()
    mask1, nfm = Frontend.msm_masks_from_decisions(dec.to(DEV), g['msm_bernoulli'].to(torch.uint8).to(DEV), T, f)
    assert torch.equal(mask1.cpu().bool(), g['msm_mask1'])
    assert torch.equal(nfm.cpu(), (strat != 2).float())


def test_motion_color_on_device(golden):
    """visual_aug_mode='motion_color' (dalle_bert.py:140-158, 940-943): the device kernel's output equals
    oracle.frontend.video_color_shift -- pinned to the reference's warp_video_with_color -- for the parameters it drew;
    frame 0 is never touched; the gate fires with probability 0.9; the BERT / ART-V forward accept the mode."""
    from mmvid_amd.frontend import Frontend
    from oracle.frontend import video_color_shift
    g = golden('frontend')
    video = g['video']  # [4, 3, 3, 16, 16]
    fe = Frontend(seed=11)
    gates, chans = [], []
    for rep in range(60):
        out, prm = fe.visual_color_jitter(video.to(DEV), 0.9, 1, want_params=True)
        prm = prm.cpu()
        gates.append(float(prm[0, 0]))
        assert (prm[:, 0] == prm[0, 0]).all()  # ONE gate per call
        assert torch.equal(out[:, 0].cpu(), video[:, 0])
        if prm[0, 0] > 0:
            ref = video_color_shift(video[:, 1:], [(float(s), int(c)) for _, s, c in prm.tolist()])
            assert torch.equal(out[:, 1:].cpu(), ref)
            chans += [int(c) for c in prm[:, 2].tolist()]
            assert (prm[:, 1] >= -0.5).all() and (prm[:, 1] < 0.5).all()
        else:
            assert torch.equal(out.cpu(), video)
        fe.advance(DEV)
    assert 0.75 <= np.mean(gates) <= 1.0 and set(chans) == {0, 1, 2, 3}
    # the golden's own parameters through the oracle are pinned on CPU (test_oracle_golden); here: model plumbing
    m = load_synth(tiny_bert(1, True), golden('bert_tiny_visual'), 17).train()
    gb = golden('bert_tiny_visual')
    vis = torch.rand(2, 1, 3, 64, 64, device=DEV)
    out = m(gb['text'].to(DEV), visual=vis, target=gb['frames'].to(DEV), return_loss=True, rel=True, vid=True,
            visual_aug_mode='motion_color')
    assert all(torch.isfinite(o) for o in out)


# -------------------------------------------------------------------------------- H7: samplers vs the reference's race run
def test_sampler_kernels_on_reference_race_trajectory(golden):
    """The reference's mask_predict with torch.multinomial as an exponential race on recorded variates
    (tests/golden/mask_predict_race.npz): the HIP kernels, fed the reference's logits / confidences and the same variates,
    take the reference's decisions -- every sampled token and its probability, every keep set, the sequential candidate
    update, the best-candidate choice and the dynamic stop."""
    from mmvid_amd import ops
    g = golden('mask_predict_race')
    for tag, c in g.meta['cases'].items():
        nv, steps, dyn, Bm = c['videos'], c['steps'], c['dynamic'], c['B']
        logits, Et, tok, Yt = g[tag + '_logits'], g[tag + '_E_tok'], g[tag + '_tok'], g[tag + '_Y_tok']
        Ek, kk, keep = g[tag + '_E_keep'], g[tag + '_k_keep'], g[tag + '_keep']
        zr, zv, final = g[tag + '_z_rel'], g[tag + '_z_vid'], g[tag + '_final']
        P, TS, V = logits.shape
        t_dev, y_dev = ops.sample_race(logits.view(P * TS, V).contiguous().to(DEV), Et.view(P * TS, V).contiguous().to(DEV))
        assert torch.equal(t_dev.cpu().view(P, TS), tok), f'{tag}: {(t_dev.cpu().view(P, TS) != tok).sum().item()} token draws differ'
        assert relerr(y_dev.cpu().view(P, TS), Yt) <= 2e-6
        ti = ki = 0
        for v in range(nv):
            Y = Yt[ti:ti + 1].clone().to(DEV).contiguous()
            I_tok = tok[ti:ti + 1].clone().to(DEV).contiguous()
            ti += 1
            Imax = I_tok.clone()
            Smax, tmax = torch.zeros(1, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
            active = torch.ones(1, dtype=torch.uint8, device=DEV)
            for t in range(1, steps):
                if not bool(active[0]):
                    break
                mask1 = ops.mp_select_keep(Y, Ek[ki:ki + Bm].view(1, Bm, TS).contiguous().to(DEV), None, int(kk[ki]))
                assert torch.equal(mask1.cpu().view(Bm, TS).bool(), keep[ki:ki + Bm]), (tag, v, t)
                ki += Bm
                pz = ti - v - 1
                ops.mp_update(mask1, Yt[ti:ti + Bm].contiguous().to(DEV), tok[ti:ti + Bm].contiguous().to(DEV),
                              zr[pz:pz + Bm].contiguous().to(DEV), zv[pz:pz + Bm].contiguous().to(DEV), t, dyn, Y, I_tok, Imax,
                              Smax, tmax, active)
                ti += Bm
            assert torch.equal(Imax.cpu()[0], final[v]), (tag, v)
        assert ti == P and ki == Ek.shape[0], (tag, ti, ki)


def test_mask_predict_free_running_against_reference_race_run(golden):
    """The whole HIP sampler (bf16 tower) on the reference's variates, free running: step 0 must reproduce the
    reference's tokens up to bf16 near-ties; the final agreement is reported (one flipped near-tie changes the inputs of
    every later step, so it is informational -- the decision-level pin is the test above)."""
    g, gb = golden('mask_predict_race'), golden('bert_tiny')
    from oracle.synth import synth_tokens
    m = load_synth(tiny_bert(), gb, 17).eval()
    text = synth_tokens('text', (2, 16), 49408, 17, low=1)
    text[0, 11:] = 0
    text[1, 5:] = 0
    c = g.meta['cases']['a']
    Et, Ek, tok = g['a_E_tok'], g['a_E_keep'], g['a_tok']
    Bm, steps, TS = c['B'], c['steps'], 32
    per_video = 1 + (steps - 1) * Bm  # tower passes per video in the reference's order: video-major
    trace = []

    def race(name, shape):
        if name == 'tok0':
            return torch.cat([Et[v * per_video] for v in range(2)]).to(DEV)
        t = int(name[3:] if name.startswith('tok') else name[4:])
        if name.startswith('keep'):
            rows = [Ek[v * (steps - 1) * Bm + (t - 1) * Bm + j] for v in range(2) for j in range(Bm)]
            return torch.stack(rows).view(2, Bm, TS).to(DEV)
        rows = [Et[v * per_video + 1 + (t - 1) * Bm + j] for v in range(2) for j in range(Bm)]
        return torch.cat(rows).to(DEV)

    mp = dict(g.meta['mp_config'], B=Bm)
    _, _, seq = m.generate_images(text.to(DEV), mask_predict_steps=steps, mp_config=mp, dynamic=False, _race=race, _trace=trace)
    step0 = trace[0]['I_tok'].cpu()
    ref0 = torch.stack([tok[v * per_video] for v in range(2)])
    agree0 = (step0 == ref0).float().mean().item()
    agree = (seq.cpu().view(2, TS) == g['a_final']).float().mean().item()
    print(f'free-running HIP sampler vs the reference race run: step-0 tokens {agree0:.3f}, final tokens {agree:.3f} equal')
    assert agree0 >= 0.9


# --------------------------------------------------------------------------------------------- X1: stable=True on the device
def test_stable_divide_max_on_device(golden):
    """BERT(stable=True): transformer_forward divides the tower output by its row maximum (utils/utils.py:18-25,
    dalle_bert.py:489-493); forward + backward run on the HIP path and agree with the unnormalised model / amax."""
    g = golden('bert_tiny')
    m = load_synth(tiny_bert(), g, 17).train()
    ms = load_synth(tiny_bert(stable=True), g, 17).train()
    assert ms.stable and ms.norm_by_max is not None
    x = torch.randn(2, ms.total_seq_len, 768, device=DEV)
    y, ys = m.transformer_forward(x), ms.transformer_forward(x)
    assert torch.allclose(ys, y / y.amax(dim=-1, keepdim=True), rtol=1e-6, atol=1e-7)
    text, frames = g['text'].to(DEV), g['frames'].to(DEV)
    lm, lr, lv = ms(text, target=frames, return_loss=True, rel=True, vid=True, _mask1=g['mask1'], _target_warp=g['warped_frames'].to(DEV))
    (7 * lm + 0.5 * lr + 0.5 * lv).backward()
    gr = ms.transformer.transformer.resblocks[0].mlp.c_fc.weight.grad
    assert all(torch.isfinite(v) for v in (lm, lr, lv)) and torch.isfinite(gr).all() and gr.abs().sum() > 0
    # oracle: the same losses with the same normalisation (fp32 CPU)
    from oracle import bert as ob
    from conftest import synth_model_sd
    sd = synth_model_sd(g, 17)
    cfg = ob.Cfg(sd, 16, 0, 2, 64)
    r = ob.forward_losses(sd, cfg, g['text'], g['target_tok'], g['mask1'], g['warp_tok'], rel_no_fully_masked=False, stable=True)
    tt = ms.get_image_tokens(frames).cpu()
    if torch.equal(tt, g['target_tok']):  # bf16 encoder reproduced the tokens: the losses are comparable
        for got, ref, nm in ((lm, r['loss_msm'], 'msm'), (lr, r['loss_rel'], 'rel'), (lv, r['loss_vid'], 'vid')):
            assert abs(got.item() - ref.item()) <= 3e-2 * max(1.0, abs(ref.item())), (nm, got.item(), ref.item())


# -------------------------------------------------------------------------------- width-512 text tower on the HIP path
def test_openai_clip_text_tower_width512():
    """which_transformer='openai_clip_text' (clip_model.py:538-547: width 512, 8 heads of 64): forward + backward against the
    fp32 oracle tower on the same weights."""
    from mmvid_amd.clip_tower import OpenAICLIPTransformer
    from oracle import tower as ot
    torch.manual_seed(3)
    L = 77
    tw = OpenAICLIPTransformer(L, 'openai_clip_text', causal=True, mask_type='causal', layers=2).to(DEV)
    assert (tw.width, tw.heads) == (512, 8)
    with torch.no_grad():
        for p in tw.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.03)
    x = torch.randn(2, L, 512, device=DEV, requires_grad=True)
    gy = torch.randn(2, L, 512, device=DEV)
    y = tw(x)
    y.backward(gy)
    sd = {k: v.detach().float().cpu().requires_grad_(True) for k, v in tw.state_dict().items()}
    xc = x.detach().cpu().requires_grad_(True)
    mask = torch.full((L, L), float('-inf')).triu_(1)
    yr = ot.tower(sd, xc, mask, 'transformer.', heads=8)
    yr.backward(gy.cpu())
    close(y, yr, 2e-2, 'width-512 tower y')
    close(x.grad, xc.grad, 3e-2, 'width-512 tower dx')
    close(tw.transformer.resblocks[0].attn.in_proj_weight.grad, sd['transformer.resblocks.0.attn.in_proj_weight'].grad, 3e-2, 'd in_proj')
    close(tw.transformer.resblocks[1].mlp.c_proj.weight.grad, sd['transformer.resblocks.1.mlp.c_proj.weight'].grad, 3e-2, 'd c_proj')


# ------------------------------------------------------------------- (e): row-wise exchange with a simulated second rank
def test_sparse_exchange_with_simulated_peer_on_device(golden):
    """FlatTrainer._exchange_sparse on the GPU with world = 2 simulated on one device: `_gather` is fed a fake peer's
    (row ids, gradient rows), so the branch that adds OTHER ranks' rows (never taken with one real rank) runs as device
    code; the result must equal the dense sum of both ranks' table gradients.  The dense `_send` path is checked the same
    way through a fake all-reduce."""
    from mmvid_amd.engine import FlatTrainer, backward_order
    g = golden('bert_tiny')
    m = load_synth(tiny_bert(), g, 17).train()
    tr = FlatTrainer(m, order=backward_order)
    text, frames = g['text'].to(DEV), g['frames'].to(DEV)
    tr.zero_grad()
    lm, lr, lv = m(text, target=frames, return_loss=True, rel=True, vid=True)
    (7 * lm + 0.5 * lr + 0.5 * lv).backward()
    W = m.text_emb.weight.grad
    mine = W.clone()
    ids = m.sparse_grad_rows()['text_emb.weight']
    assert ids is not None and tr._sparse_ranges(), 'the text embedding is exchanged row-wise after a logged forward'
    # the fake peer: overlapping, repeated and disjoint row ids, random rows; its dense table gradient
    torch.manual_seed(1)
    n = ids.numel()
    peer_ids = torch.cat((ids[:n // 2], torch.randint(0, 49408, (n - n // 2, ), device=DEV)))
    peer_dense = torch.zeros_like(W)
    peer_dense[peer_ids.unique()] = torch.randn(peer_ids.unique().numel(), W.shape[1], device=DEV)
    for my_rank in (0, 1):
        W.copy_(mine)
        puid, prows = FlatTrainer.pack_rows(peer_dense, peer_ids)
        calls = []

        def fake_gather(t, my_rank=my_rank, puid=puid, prows=prows):
            peer = puid if t.dim() == 1 else prows
            assert peer.shape == t.shape and peer.dtype == t.dtype
            calls.append(t.shape)
            return torch.cat((t, peer) if my_rank == 0 else (peer, t), 0)

        tr.world, tr._gather, tr._rank = 2, fake_gather, (lambda my_rank=my_rank: my_rank)
        tr._exchange_sparse()
        assert len(calls) == 2
        torch.testing.assert_close(W, mine + peer_dense, rtol=1e-6, atol=1e-6)
        if tr._lazy is not None:  # the peer's rows carry a gradient here now: Adam must not skip them
            assert bool(tr._lazy['flags'][peer_ids].all())
        touched = torch.zeros(W.shape[0], dtype=torch.bool, device=DEV)
        touched[ids] = True
        touched[peer_ids] = True
        assert W[~touched].abs().sum() == 0
    # no forward logged (or the log overflowed) -> the table is NOT cut out of the dense all-reduce
    tr.zero_grad()
    assert m.sparse_grad_rows()['text_emb.weight'] is None and tr._sparse_ranges() == []
    with torch.no_grad():
        for _ in range(m.TEXT_ID_LOG_MAX + 1):
            m._log_text_ids(torch.zeros(2, 1 + 16, dtype=torch.long, device=DEV))
    assert m.sparse_grad_rows()['text_emb.weight'] is None  # torch.no_grad(): nothing is logged
    for _ in range(m.TEXT_ID_LOG_MAX + 1):
        m._log_text_ids(torch.zeros(2, 1 + 16, dtype=torch.long, device=DEV))
    assert m._text_id_overflow and m.sparse_grad_rows()['text_emb.weight'] is None and tr._sparse_ranges() == []


def test_frontend_seed_lives_on_the_device_and_follows_torch_seed(golden):
    """the front-end seed defaults to torch.initial_seed() (+ rank), is read by the kernels from device memory (a
    captured step follows `frontend.seed = ...`), and (seed, step) travel with the trainer's state_dict."""
    from mmvid_amd.frontend import Frontend
    torch.manual_seed(1234)
    fe = Frontend()
    a = fe.msm_masks(8, 2, 4, DEV, [0.25] * 4, [0.2, 0.5])[0].clone()
    assert fe.seed == 1234
    fe2 = Frontend(seed=1234)
    assert torch.equal(a, fe2.msm_masks(8, 2, 4, DEV, [0.25] * 4, [0.2, 0.5])[0])
    graph = torch.cuda.CUDAGraph()
    out = torch.empty(64, 32, device=DEV, dtype=torch.uint8)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fe2.msm_masks(64, 2, 4, DEV, [0.25] * 4, [0.2, 0.5])
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        out.copy_(fe2.msm_masks(64, 2, 4, DEV, [0.25] * 4, [0.2, 0.5])[0])
    graph.replay()
    first = out.clone()
    fe2.seed = 99  # written into the device state: the captured kernel reads it on the next replay
    graph.replay()
    assert not torch.equal(first, out)
    fe3 = Frontend(seed=99)
    assert torch.equal(out, fe3.msm_masks(64, 2, 4, DEV, [0.25] * 4, [0.2, 0.5])[0])
    fe3.advance(DEV), fe3.advance(DEV)
    sd = fe3.state_dict()
    assert sd == {'seed': 99, 'step': 2.0}
    fe4 = Frontend()
    fe4.load_state_dict(sd)
    assert torch.equal(fe4.msm_masks(64, 2, 4, DEV, [0.25] * 4, [0.2, 0.5])[0], fe3.msm_masks(64, 2, 4, DEV, [0.25] * 4, [0.2, 0.5])[0])


def test_lazy_table_rows_are_exact(golden):
    """FlatTrainer(lazy_rows=True): rows of text_emb that never received a gradient are skipped by Adam, the gradient norm
    and zero_grad.  Must be EXACT: parameters, moments and the bf16 shadow equal the dense trainer's bit for bit over
    several steps with changing text (new rows get touched, old ones keep decaying), and across a state_dict round trip."""
    from mmvid_amd.engine import FlatTrainer, backward_order
    g = golden('bert_tiny')
    frames = g['frames'].to(DEV)
    ms = [load_synth(tiny_bert(), g, 17).train() for _ in range(2)]
    trs = [FlatTrainer(ms[0], lr=1e-3, order=backward_order, lazy_rows=True), FlatTrainer(ms[1], lr=1e-3, order=backward_order, lazy_rows=False)]
    assert trs[0]._lazy is not None and trs[1]._lazy is None
    gen = torch.Generator().manual_seed(3)
    touched = set()
    for step in range(5):
        text = torch.randint(1, 49408, (2, 16), generator=gen)
        text[0, 9:] = 0
        touched |= set(torch.where(text == 0, torch.arange(16) + 49408, text).view(-1).tolist())
        mask1 = torch.rand(2, 32, generator=gen) < 0.4
        # one backward (the scatter-add / bias column sums use fp32 atomics: two backward passes differ in the last bits), the
        # SAME gradient buffer through both optimisers
        trs[0].zero_grad(), trs[1].zero_grad()
        lm, lr, lv = ms[0](text.to(DEV), target=frames, return_loss=True, rel=True, vid=True, _mask1=mask1, _target_warp=frames)
        (7 * lm + 0.5 * lr + 0.5 * lv).backward()
        trs[1].G.copy_(trs[0].G)
        trs[0].step(), trs[1].step()
        assert torch.equal(trs[0].P, trs[1].P), f'step {step}: parameters differ'
        assert torch.equal(trs[0].M, trs[1].M) and torch.equal(trs[0].V, trs[1].V) and torch.equal(trs[0].S, trs[1].S)
        assert float(trs[0]._sq) == float(trs[1]._sq)
    flags = trs[0]._lazy['flags'].cpu()
    assert set(flags.nonzero().view(-1).tolist()) >= touched and int(flags.sum()) <= len(touched) + 1
    # the table gradient outside the touched rows was never written and stays zero; zero_grad clears what the step left
    trs[0].zero_grad()
    assert float(trs[0].G.abs().sum()) == 0.0
    # a loaded optimiser state raises the flags of every row with live moments
    tr2 = FlatTrainer(load_synth(tiny_bert(), g, 17).train(), lr=1e-3, order=backward_order)
    tr2.load_state_dict(trs[1].state_dict())
    assert int(tr2._lazy['flags'].sum()) >= len(touched)


# --------------------------------------------------------------------------- the loader-wave GEMM block on ragged shapes
@pytest.mark.parametrize('M,N,K', [(10422, 2304, 768), (10422, 768, 3072), (2561, 776, 200), (777, 2304, 128), (300, 3072, 768),
                                   (70000, 256, 64)])
def test_gemm_loader_wave_block_epilogues_vs_torch(M, N, K):
    """The 256x128 loader-wave GEMM block (8 MFMA waves + 4 loader waves, register-direct epilogues, column-group tile order) on ragged
    M / N / K edges, a single tile and several tiles per persistent block: packed bf16 with bias, fp32 with bias, QuickGELU + saved
    pre-activation, and the dX form with QuickGELU' and column sums -- against fp32 torch on the bf16 operands.  (Rounds 3-4 held the
    LDS-staged, register-direct, eight-loader and four-fat-wave forms bit-identical to each other; only this form is left.)"""
    from mmvid_amd import ops
    torch.manual_seed(M + N + K)
    bf = torch.bfloat16
    A = torch.randn(M, K, device=DEV).to(bf)
    W = (torch.randn(N, K, device=DEV) * 0.05).to(bf)
    Wk = (torch.randn(K, N, device=DEV) * 0.05).to(bf)
    bias = torch.randn(N, device=DEV) * 0.1
    pre_in = torch.randn(M, N, device=DEV).to(bf)
    save = torch.zeros(M, N, device=DEV, dtype=bf)
    cs = torch.zeros(N, device=DEV)
    o16 = ops.gemm(A, W, bias=bias)                                      # qkv-like: packed bf16
    o32 = ops.gemm(A, W, bias=bias, out_dtype=torch.float32)             # fp32 result
    act = ops.gemm(A, W, bias=bias, act=1, save_pre=save)                # c_fc-like: two bf16 results
    dx = ops.gemm(A, Wk, b_kmajor=True, dact_pre=pre_in, colsum=cs)      # d_pre-like
    want = A.float() @ W.float().t() + bias
    assert relerr(o32.cpu(), want.cpu()) < 1e-5
    assert torch.equal(o16, want.to(bf)) or relerr(o16.float().cpu(), want.cpu()) < 4e-3
    assert torch.equal(save, o16), 'the saved pre-activation is the bf16 result before the activation'
    assert relerr(act.float().cpu(), (want * torch.sigmoid(1.702 * want)).cpu()) < 6e-3
    p32 = pre_in.float()
    sg = torch.sigmoid(1.702 * p32)
    wdx = (A.float() @ Wk.float()) * (sg * (1 + 1.702 * p32 * (1 - sg)))
    assert relerr(dx.float().cpu(), wdx.cpu()) < 6e-3
    assert relerr(cs.cpu(), wdx.sum(0).cpu()) < 2e-3  # (unrounded fp32 sums of what is stored)
    again = ops.gemm(A, W, bias=bias)
    assert torch.equal(again, o16), 'not deterministic'


def test_attention_backward_with_fused_in_proj_bias_gradient():
    """mmvid_attention_bwd_bias: dqkv as mmvid_attention_bwd, and dbias = column sums of dqkv (nn.MultiheadAttention
    in_proj_bias gradient) taken from the kernels' registers; against fp32 torch autograd, with padded rows (L = 579)."""
    from mmvid_amd import _lib, ops
    B, L, H, E = 3, 579, 12, 768
    torch.manual_seed(0)
    qkv = (torch.randn(B * L, 3 * E, device=DEV) * 0.5).bfloat16()
    dO = (torch.randn(B * L, E, device=DEV) * 0.1).bfloat16()
    out = torch.empty(B * L, E, device=DEV, dtype=torch.bfloat16)
    lse, delta = torch.empty(B * H * L, device=DEV), torch.empty(B * H * L, device=DEV)
    dqkv = torch.empty(B * L, 3 * E, device=DEV, dtype=torch.bfloat16)
    db = torch.zeros(3 * E, device=DEV)
    st = ops._stream
    _lib.call('mmvid_attention_fwd', ops._p(qkv), 3 * E, B, L, H, E, 0.125, 2, 65, 65, 66, 66, ops._p(out), E, ops._p(lse), st())
    _lib.call('mmvid_attention_bwd_bias', ops._p(qkv), 3 * E, ops._p(out), E, ops._p(dO), E, ops._p(lse), ops._p(delta), B, L, H, E, 0.125,
              2, 65, 65, 66, 66, ops._p(dqkv), 3 * E, ops._p(db), st())
    q, k, v = [t.float().view(B, L, H, 64).transpose(1, 2).requires_grad_(True) for t in qkv.float().split(E, dim=1)]
    mask = torch.zeros(L, L, device=DEV)
    mask[65, :65] = float('-inf')
    mask[66, :66] = float('-inf')
    o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + mask, -1) @ v).transpose(1, 2).reshape(B * L, E)
    o.backward(dO.float())
    g = torch.cat([t.grad.transpose(1, 2).reshape(B * L, E) for t in (q, k, v)], 1)
    close(out, o, 2e-2, 'attention out')
    close(dqkv, g, 3e-2, 'dqkv')
    close(db, g.sum(0), 1e-2, 'fused in_proj bias gradient')


def test_tower_12_layers_at_training_length_vs_reference(golden):
    """The HIP tower at full depth and the training shape (12 layers, L = 579, restricted rows 65 / 66) against the reference's own
    forward + backward (tests/golden/tower12.npz): bf16 tolerances after 12 layers, norms within 2 %."""
    from mmvid_amd.clip_tower import OpenAICLIPTransformer
    from oracle.synth import synth_input
    g = golden('tower12')
    L = g.meta['L']
    tw = load_synth(OpenAICLIPTransformer(L, 'openai_clip_visual', causal=True, mask_type='mask_prev', mask_kwargs={'index': [65, 66]},
                                          layers=12), g, 23)
    x = synth_input('x_t12', (2, L, 768), 23).to(DEV).requires_grad_(True)
    gy = synth_input('g_t12', (2, L, 768), 23).to(DEV)
    y = tw(x)
    y.backward(gy)
    rows = [0, 64, 65, 66, 67, 578]
    close(y[:, ::37, ::13], g['y_s'], 3e-2, '12-layer tower y')
    close(x.grad[:, ::37, ::13], g['dx_s'], 5e-2, '12-layer tower dx')
    close(y[:, rows][..., ::7], g['y_rows'], 3e-2, '12-layer tower y, rows around the restricted ones')
    close(x.grad[:, rows][..., ::7], g['dx_rows'], 5e-2, '12-layer tower dx, rows around the restricted ones')
    assert abs(y.double().norm().item() / g['y_norm'].item() - 1) < 2e-2
    assert abs(x.grad.double().norm().item() / g['dx_norm'].item() - 1) < 2e-2
    blk = tw.transformer.resblocks
    for li in (0, 5, 11):
        for nm, prm in (('inw', blk[li].attn.in_proj_weight), ('outw', blk[li].attn.out_proj.weight), ('fcw', blk[li].mlp.c_fc.weight),
                        ('pjw', blk[li].mlp.c_proj.weight)):
            close(prm.grad[::61, ::29], g[f'l{li}_d{nm}_s'], 5e-2, f'layer {li} d{nm}')
            assert abs(prm.grad.double().norm().item() / g[f'l{li}_d{nm}_norm'].item() - 1) < 2e-2
        for nm, prm in (('inb', blk[li].attn.in_proj_bias), ('ln1w', blk[li].ln_1.weight), ('ln2b', blk[li].ln_2.bias),
                        ('fcb', blk[li].mlp.c_fc.bias), ('pjb', blk[li].mlp.c_proj.bias)):
            close(prm.grad, g[f'l{li}_d{nm}'], 5e-2, f'layer {li} d{nm}')


@pytest.mark.parametrize('B', [1, 4, 8])
def test_decode_step_on_a_long_cache(B):
    """Decode steps deep into a long cache (position 643..647 of 704): the gemv rows live in LDS as bf16 and are reduced by the DPP
    reduce-scatter (csrc/decode.hip), for 1 / 4 / 8 rows -- against the full-prefix causal forward at the same positions."""
    from mmvid_amd.clip_tower import OpenAICLIPTransformer
    torch.manual_seed(B)
    L, P, steps = 704, 643, 5
    tw = OpenAICLIPTransformer(seq_len=L, which_model='openai_clip_visual', causal=True, layers=2).to(DEV).eval()
    x = torch.randn(B, P + steps, 768, device=DEV) * 0.5
    with torch.no_grad():
        full = tw(x)
        cache = tw.new_kv_cache(B, L, DEV)
        tw.prefill(x[:, :P].contiguous(), cache)
        sess = tw.decode_session(cache, P)
        for k in range(steps):
            h = sess.step(x[:, P + k].contiguous())
            close(h, full[:, P + k], 1e-2, f'B={B}: incremental vs full forward at position {P + k}')


def test_dpp_wave_reductions():
    """wave_sum_fast / wave_max_fast (csrc/common.h: DPP inside a row, v_permlane16_swap / v_permlane32_swap across rows): every lane
    holds the total; equal to the ds_bpermute butterfly up to fp32 summation order."""
    from mmvid_amd import _lib, ops
    torch.manual_seed(3)
    for scale in (1.0, 1e3):
        v = (torch.randn(64, device=DEV) * scale).contiguous()
        out = torch.empty(3, 64, device=DEV)
        _lib.call('mmvid_probe', 3, ops._p(v), ops._p(out), ops._stream())
        ref = v.double().sum().item()
        assert torch.all(out[0] == out[0, 0]) and torch.all(out[1] == out[1, 0])
        assert abs(out[0, 0].item() - ref) <= 1e-5 * v.abs().sum().item()
        assert out[1, 0].item() == v.max().item()
        assert abs(out[2, 0].item() - ref) <= 1e-5 * v.abs().sum().item()
    # exact on integers: the order of the additions cannot matter
    v = torch.arange(64, device=DEV, dtype=torch.float32)
    _lib.call('mmvid_probe', 3, ops._p(v), ops._p(out), ops._stream())
    assert torch.all(out[0] == 2016.0) and torch.all(out[1] == 63.0)


# ------------------------------------------------------------------------------------------- vae.strict = 'split'
def _conv_ref64(x, w, b, mode, residual=None):
    """fp64 reference on the CPU: x [N,H,W,Cin], w [Cout,taps,Cin] -> [N,Ho,Wo,Cout] (modes of mmvid_conv2d_nhwc)."""
    import torch.nn.functional as F
    xd = x.double().cpu().permute(0, 3, 1, 2)
    cout, taps, cin = w.shape
    k = 3 if taps == 9 else 1
    wd = w.double().cpu().view(cout, k, k, cin).permute(0, 3, 1, 2)
    if mode == 1:
        y = F.conv2d(F.pad(xd, (0, 1, 0, 1)), wd, b.double().cpu(), stride=2)
    elif mode == 2:
        y = F.conv2d(F.interpolate(xd, scale_factor=2.0, mode='nearest'), wd, b.double().cpu(), padding=1)
    else:
        y = F.conv2d(xd, wd, b.double().cpu(), padding=k // 2)
    y = y.permute(0, 2, 3, 1)
    return y + residual.double().cpu() if residual is not None else y


@pytest.mark.parametrize('mode,n,h,cin,cout,strip,splitk', [
    (0, 2, 32, 128, 128, True, 1),    # strip form (ResnetBlock convs at 32x32 and above)
    (0, 3, 16, 256, 256, False, 1),   # implicit GEMM, fast A path
    (0, 2, 8, 512, 512, False, 4),    # deep layer on a small map: split-K
    (1, 2, 32, 128, 128, False, 1),   # Downsample
    (2, 2, 8, 64, 64, False, 1),      # Upsample (decoder)
    (3, 2, 16, 256, 256, False, 1),   # 1x1
    (0, 2, 32, 8, 128, False, 1),     # conv_in: 3 real channels padded to 8
])
def test_split_convolution_vs_fp64(mode, n, h, cin, cout, strip, splitk):
    """mmvid_conv2d_nhwc_split3 / mmvid_conv3x3_strip_nhwc_split3: x_hi.w_hi + x_lo.w_hi + x_hi.w_lo in one K loop.  Against an
    fp64 convolution of the fp32 operands: the error must be what 16 mantissa bits per operand allow (~1e-5 of the output
    scale), three orders of magnitude below the bf16 operator's."""
    from mmvid_amd import ops
    torch.manual_seed(mode * 7 + cin)
    taps = 1 if mode == 3 else 9
    x = torch.randn(n, h, h, cin, device=DEV)
    if cin == 8:
        x[..., 3:] = 0
    w = torch.randn(cout, taps, cin, device=DEV) / math.sqrt(taps * cin)
    b = torch.randn(cout, device=DEV)
    ho = h // 2 if mode == 1 else (2 * h if mode == 2 else h)
    res = torch.randn(n, ho, ho, cout, device=DEV)
    y = ops.conv2d_nhwc_split3(ops.split_planes(x), ops.split_weights(w), b, mode, residual=res, splitk=splitk, strip=strip)
    ref = _conv_ref64(x, w, b, mode, res)
    err = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    y16 = ops.conv2d_nhwc(x.to(torch.bfloat16), w.to(torch.bfloat16), b, mode, residual=res, out_dtype=torch.float32)
    err16 = (y16.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f'split conv mode {mode} {h}x{h} {cin}->{cout}: max err / max |y| = {err:.2e} (bf16 operator: {err16:.2e})')
    assert err < 1e-5 and err16 > 50 * err
    # pair planes: hi + lo carries 16 mantissa bits of x
    pl = ops.split_planes(x)
    assert ((pl[0].float() + pl[1].float() - x).abs() <= x.abs() * 2.0**-16).all()


def test_split_groupnorm_vs_fp64():
    import torch.nn.functional as F
    from mmvid_amd import ops
    torch.manual_seed(5)
    for (n, h, c, swish) in ((2, 32, 128, True), (3, 16, 256, False), (2, 8, 512, True)):
        x = torch.randn(n, h, h, c, device=DEV) * 2 + 0.7
        w, b = torch.randn(c, device=DEV), torch.randn(c, device=DEV)
        pl = ops.groupnorm_swish_split(x, w, b, swish=swish)
        ref = F.group_norm(x.double().cpu().permute(0, 3, 1, 2), 32, w.double().cpu(), b.double().cpu(), 1e-6)
        if swish:
            ref = ref * torch.sigmoid(ref)
        ref = ref.permute(0, 2, 3, 1)
        got = pl[0].double().cpu() + pl[1].double().cpu()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        print(f'split GroupNorm {h}x{h}x{c} swish={swish}: max err / max |y| = {err:.2e}')
        assert err < 2e-5


@pytest.mark.parametrize('n,h,cin,cout', [(2, 128, 128, 128), (3, 64, 128, 128), (1, 32, 256, 256), (2, 64, 32, 128)])
def test_fp16_single_product_convolution_and_groupnorm_vs_fp64(n, h, cin, cout):
    """The fp16 layer forms of vae.strict = 'mixed': GroupNorm + swish written as one fp16 plane (the fp32 result rounded to nearest
    even: bit-exact against torch's .half()), and the 3x3 strip convolution as ONE product of fp16 operands with fp32 accumulation --
    against an fp64 convolution of the SAME fp16-rounded operands (what is left is the fp32 accumulation: 1e-6), and against the
    unrounded operands (the format's own 2^-12 per operand), with a residual."""
    import torch.nn.functional as F
    from mmvid_amd import ops
    torch.manual_seed(n + h + cin)
    x = torch.randn(n, h, h, cin, device=DEV) * 1.5 + 0.2
    gw, gb = torch.randn(cin, device=DEV) * 0.1 + 1, torch.randn(cin, device=DEV) * 0.1
    y16 = ops.groupnorm_swish_f16(x, gw, gb)
    pair = ops.groupnorm_swish_split(x, gw, gb)
    y32 = pair[0].double() + pair[1].double()  # the same arithmetic to 2^-17
    assert y16.dtype == torch.float16 and ((y16.double() - y32).abs() <= 2.0**-11 * y32.abs() + 1e-7).all()
    ref = F.group_norm(x.double().permute(0, 3, 1, 2), 32, gw.double(), gb.double(), 1e-6)
    ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1)
    assert ((y16.double() - ref).abs().max() / ref.abs().max()).item() < 1e-3
    w = torch.randn(cout, 9, cin, device=DEV) / (9 * cin)**0.5
    b = torch.randn(cout, device=DEV) * 0.02
    res = torch.randn(n, h, h, cout, device=DEV)
    w16 = w.half()
    out = ops.conv3x3_strip_f16(y16, w16, b, residual=res)

    def conv64(a, wt):
        wt4 = wt.double().view(cout, 3, 3, cin).permute(0, 3, 1, 2)
        return (F.conv2d(a.double().permute(0, 3, 1, 2), wt4, b.double(), padding=1).permute(0, 2, 3, 1) + res.double())
    same = conv64(y16, w16)
    e_same = ((out.double() - same).abs().max() / same.abs().max()).item()
    full = conv64(y32, w)
    e_full = ((out.double() - full).abs().max() / full.abs().max()).item()
    print(f'fp16 conv {n}x{h}x{h} {cin}->{cout}: vs fp64 on the fp16 operands {e_same:.2e}, vs fp64 on the unrounded operands {e_full:.2e}')
    assert e_same < 3e-6 and e_full < 1.5e-3
    assert torch.equal(out, ops.conv3x3_strip_f16(y16, w16, b, residual=res))
    # the same result stored as a bf16 pair by the epilogue (what the encoder's last convolution of a level hands its Downsample): the
    # planes mmvid_split_f32_bf16x2 makes of the fp32 result, bit for bit -- for this kernel and for the three-product pair operator
    planes = torch.empty(2, n, h, h, cout, device=DEV, dtype=torch.bfloat16)
    out2 = ops.conv3x3_strip_f16(y16, w16, b, residual=res, planes_out=planes)
    assert torch.equal(out2, out) and torch.equal(planes, ops.split_planes(out))
    w3 = ops.split_weights(w)
    o3 = ops.conv2d_nhwc_split3(pair, w3, b, 0, residual=res, strip=True)
    planes.zero_()
    o3b = ops.conv2d_nhwc_split3(pair, w3, b, 0, residual=res, strip=True, planes_out=planes)
    assert torch.equal(o3b, o3) and torch.equal(planes, ops.split_planes(o3))


@pytest.mark.parametrize('mode', ['split', 'mixed'])
@pytest.mark.parametrize('name,tiny', [('vqgan_tiny', True), ('vqgan_full', False)])
def test_split_encoder_indices_equal_reference(golden, name, tiny, mode):
    """vae.strict = 'split' (bf16-pair convolutions on the bf16 matrix pipe): the indices equal the reference's on the VQGAN
    goldens, z_e / decode agree to ~1e-4 (between the bf16 operator's 3e-2 and the fp32 operator's 4e-6).  'mixed' (round 5: the 3x3
    residual-block convolutions of the encoder's 128x128 / 64x64 levels as one fp16 product): the same indices, z_e to 4e-3; its
    decoder is the pair operator's."""
    from mmvid_amd.vae import VQGanVAE1024
    from oracle.synth import synth_input
    from test_host_logic import tiny_vae
    from test_models_gpu import close, load_synth
    g = golden(name)
    s = g.meta['image_size']
    vae = tiny_vae() if tiny else VQGanVAE1024(None, 128)
    vae.image_size = s
    load_synth(vae, g, 11)
    vae.strict = mode
    img = synth_input('img', (g.meta['n'], 3, s, s), 11, 'uniform').to(DEV)
    z = vae.encode_z(img)
    ref = g['z_e'].permute(0, 2, 3, 1)
    print(f"{name} {mode} z_e: max |dz| {(z.cpu() - ref).abs().max().item():.3e} (max |z| {ref.abs().max().item():.2f})")
    close(z, ref, 2e-4 if mode == 'split' else 4e-3, f'{name} {mode} z_e')
    idx = vae.get_codebook_indices(img).cpu()
    assert torch.equal(idx, g['indices']), f'{name}: {(idx != g["indices"]).sum().item()} of {idx.numel()} indices differ'
    dec = vae.decode(g['indices'].to(DEV))
    print(f"{name} split decode: max |d| {(dec.cpu() - g['decoded']).abs().max().item():.3e}")
    close(dec, g['decoded'], 2e-4, f'{name} split decode')
    assert torch.equal(vae.get_codebook_indices(img[:1]).cpu(), idx[:1])  # batch composition does not matter
    vae.strict = False
    assert vae.get_codebook_indices(img).shape == idx.shape


@pytest.mark.parametrize('mode', ['split', 'mixed'])
@pytest.mark.parametrize('name,nv,cvae', [('bert_tiny', 0, False), ('bert_tiny_visual', 1, True)])
def test_split_tokens_of_bert_goldens(golden, name, nv, cvae, mode):
    from test_host_logic import tiny_bert
    from test_models_gpu import load_synth
    g = golden(name)
    m = load_synth(tiny_bert(nv, cvae), g, 17)
    m.vae.strict = mode
    if m.cvae is not None:
        m.cvae.strict = mode
    assert torch.equal(m.get_image_tokens(g['frames'].to(DEV)).cpu(), g['target_tok'])
    assert torch.equal(m.get_image_tokens(g['warped_frames'].to(DEV)).cpu(), g['warp_tok'])
    if nv:
        assert torch.equal(m.get_image_tokens(g['visual'].to(DEV), which_vae='cvae').cpu(), g['visual_tok'])


# ------------------------------------------------------------------------------ BERT with a fixed language model
@pytest.mark.parametrize('name,bn', [('bert_flm', None), ('bert_flm_bottleneck', '256')])
def test_bert_fixed_language_model_vs_reference(golden, name, bn):
    """dalle_bert.py:307-322, 924-925 on the HIP path: the sentence feature goes through text_feature_mapping (MFMA GEMMs +
    LayerNorm kernels) into the one text token; REL swaps it with the rest of the control.  Losses / gradients vs the
    reference's with its tokens, mask and warped tokens injected; tokens of the frames exact in the pair-operator mode."""
    from mmvid_amd.engine import FlatTrainer, backward_order
    g = golden(name)
    m = load_synth(tiny_bert(fixed_language_model='roberta-large', text_feature_dim=1024, text_emb_bottleneck=bn), g, 23).train()
    feat = g['text_feat'].to(DEV)
    with torch.no_grad():
        ctrl = m(feat, return_loss=False)
    close(ctrl, g['control_emb'], 1e-2, 'control_emb (text row through the bf16 mapping)')
    close(torch.cat((ctrl[:, :1], ctrl[:, 2:]), 1), torch.cat((g['control_emb'][:, :1], g['control_emb'][:, 2:]), 1), 1e-6,
          'control_emb without the text row')
    m.vae.strict = 'split'
    assert torch.equal(m.get_image_tokens(g['frames'].to(DEV)).cpu(), g['target_tok'])
    m.vae.strict = False

    def run(model):
        return model(feat, target=g['target_tok'].to(DEV), return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                     _mask1=g['mask1'], _target_warp=g['warp_tok'])
    lm, lr, lv = run(m)
    losses = torch.stack([lm, lr, lv]).detach().cpu()
    print('losses', losses.tolist(), 'ref', g['losses'].tolist())
    assert torch.allclose(losses, g['losses'], rtol=2e-2, atol=2e-2)
    (7 * lm + 0.5 * lr + 0.5 * lv).backward()
    G = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    n = 0
    for k, v in G.items():
        if k.startswith('text_feature_mapping.'):
            close(v if v.dim() == 1 else v[::4, ::8], g['g_' + k], 6e-2, 'g ' + k)
            assert abs(v.double().norm().item() / g['gn_' + k].item() - 1) < 5e-2, k
            n += 1
    assert n == (2 if bn is None else 10)
    close(G['image_emb.weight'][::3, ::5], g['g_image_emb'], 5e-2, 'g image_emb')
    close(G['special_emb.weight'], g['g_special_emb'], 5e-2, 'g special_emb')
    tn = torch.sqrt(sum((v.double()**2).sum() for v in G.values())).item()
    print('total grad norm', tn, 'ref', g['g_total_norm'].item())
    assert abs(tn / g['g_total_norm'].item() - 1) < 3e-2
    # the flat engine: the mapping's Linear weights are read through bf16 shadows the fused optimiser keeps current
    m2 = load_synth(tiny_bert(fixed_language_model='roberta-large', text_feature_dim=1024, text_emb_bottleneck=bn), g, 23).train()
    tr = FlatTrainer(m2, lr=2e-5, order=backward_order)
    first = None
    for it in range(3):
        tr.zero_grad()
        a, b, c = run(m2)
        (7 * a + 0.5 * b + 0.5 * c).backward()
        tr.step()
        first = first if first is not None else float(a.detach())
    assert abs(first - float(lm.detach())) < 1e-6 * max(1.0, abs(first)) and float(a.detach()) < first, (first, float(a.detach()))
    for lin in m2.head_shadow_targets():
        assert torch.equal(m2._w16(lin), lin.weight.detach().to(torch.bfloat16)), 'stale bf16 shadow'
    m2.eval()
    torch.manual_seed(3)
    images, _, seq = m2.generate_images(feat[:1], mask_predict_steps=3, mp_config=golden('mask_predict').meta['mp_config'], dynamic=False)
    assert images.shape == (1, 2, 3, 64, 64) and seq.shape == (2, 16) and torch.isfinite(images).all()


# ------------------------------------------------------------------------------ grouped weight gradients
def test_grouped_weight_gradient_gemm():
    """mmvid_gemm_bf16_dw_grouped: G weight gradients in one launch without split-K vs the per-layer split-K GEMM and an fp64
    product; a skipped (null) entry, more groups than one launch holds, accumulate on / off, ragged tiles."""
    from mmvid_amd import ops
    torch.manual_seed(0)
    for G, M, N, K in ((18, 1000, 768, 256), (3, 2317, 264, 776), (12, 579, 2304, 768)):
        dY = (torch.randn(G, M, N, device=DEV) * 0.1).to(torch.bfloat16)
        X = (torch.randn(G, M, K, device=DEV) * 0.1).to(torch.bfloat16)
        ref = torch.einsum('gmn,gmk->gnk', dY.double(), X.double())
        base = torch.randn(G, N, K, device=DEV)
        for acc in (False, True):
            outs = [base[g].clone() for g in range(G)]
            skip = G // 2
            outs_arg = [o if g != skip else None for g, o in enumerate(outs)]
            ops.gemm_dw_grouped(dY, X, outs_arg, accumulate=acc)
            per = [base[g].clone() for g in range(G)]
            for g in range(G):
                ops.gemm_dw(dY[g], X[g], per[g], accumulate=acc)
            for g in range(G):
                if g == skip:
                    assert torch.equal(outs[g], base[g]), 'a null entry must leave its output alone'
                    continue
                want = ref[g] + (base[g].double() if acc else 0)
                e = ((outs[g].double() - want).abs().max() / want.abs().max()).item()
                e2 = ((outs[g] - per[g]).abs().max() / want.abs().max()).item()
                assert e < 2e-5 and e2 < 2e-5, (G, M, N, K, acc, g, e, e2)
            again = [base[g].clone() if g != skip else None for g in range(G)]
            ops.gemm_dw_grouped(dY, X, again, accumulate=acc)
            assert all(torch.equal(a, o) for a, o in zip(again, outs_arg) if a is not None), 'not deterministic'


def test_multi_shape_grouped_weight_gradient_gemm():
    """mmvid_gemm_bf16_dw_multi: the four Linear shapes of a ResidualAttentionBlock x several layers in ONE launch (what the tower
    backward issues after its layer loop), against fp64 products; a frozen kind (all outputs null), a frozen layer, 13 layers
    (more than one launch's pointer table holds for four kinds), ragged shapes."""
    from mmvid_amd import ops
    torch.manual_seed(1)
    for G, M, shapes in ((13, 700, [(256, 1024), (1024, 256), (256, 256), (768, 256)]), (3, 1237, [(264, 520), (520, 264)]),
                         (12, 579, [(768, 3072), (3072, 768), (768, 768), (2304, 768)])):
        kinds, refs, bases = [], [], []
        for ki, (N, K) in enumerate(shapes):
            dY = (torch.randn(G, M, N, device=DEV) * 0.1).to(torch.bfloat16)
            X = (torch.randn(G, M, K, device=DEV) * 0.1).to(torch.bfloat16)
            base = torch.randn(G, N, K, device=DEV)
            outs = [base[g].clone() for g in range(G)]
            frozen_kind = ki == 2 and len(shapes) == 4 and G == 13
            arg = [None if (frozen_kind or (g == 1 and ki == 0)) else o for g, o in enumerate(outs)]
            kinds.append((dY, X, arg))
            refs.append(torch.einsum('gmn,gmk->gnk', dY.double(), X.double()))
            bases.append((base, outs))
        ops.gemm_dw_multi(kinds, accumulate=True)
        for (dY, X, arg), ref, (base, outs) in zip(kinds, refs, bases):
            for g in range(G):
                if arg[g] is None:
                    assert torch.equal(outs[g], base[g])
                    continue
                want = ref[g] + base[g].double()
                e = ((outs[g].double() - want).abs().max() / want.abs().max()).item()
                assert e < 2e-5, (G, M, tuple(dY.shape), g, e)
        first = [[o.clone() if o is not None else None for o in arg] for _, _, arg in kinds]
        for (dY, X, arg), (base, outs) in zip(kinds, bases):
            for g, o in enumerate(arg):
                if o is not None:
                    o.copy_(base[g])
        ops.gemm_dw_multi(kinds, accumulate=True)
        for (_, _, arg), f in zip(kinds, first):
            assert all(torch.equal(a, b) for a, b in zip(arg, f) if a is not None), 'not deterministic'


def test_tower_backward_chunked_calls_match_one_call():
    """The tower backward keeps every layer's dY and computes the weight gradients of all layers of a CALL in one launch after its
    layer loop.  The multi-GPU engine calls it in chunks (3 layers per call here) so that finished layers can be exchanged early: the
    input gradient is bit-identical (nothing on that chain depends on the chunking), every parameter gradient agrees to fp32
    summation order, and both agree with torch autograd on the same module arithmetic through the golden tests of test_parity_gpu.py."""
    from mmvid_amd.clip_tower import OpenAICLIPTransformer
    from oracle.synth import synth_input
    L = 579
    res = {}
    for chunked in (False, True):
        torch.manual_seed(5)
        tw = OpenAICLIPTransformer(L, 'openai_clip_visual', causal=True, mask_type='mask_prev', mask_kwargs={'index': [65, 66]}, layers=6).to(DEV)
        done = []
        if chunked:
            tw.on_layers_done = done.append
        x = synth_input('x', (4, L, 768), 13).to(DEV).requires_grad_(True)
        y = tw(x)
        y.backward(synth_input('gy', (4, L, 768), 14).to(DEV))
        res[chunked] = (x.grad.clone(), {k: p.grad.clone() for k, p in tw.named_parameters()})
        if chunked:
            assert done == [3, 0]
    assert torch.equal(res[False][0], res[True][0]), 'the input gradient must not depend on where the weight gradients are computed'
    worst = 0.0
    for k, a in res[False][1].items():
        b = res[True][1][k]
        e = ((a - b).abs().max() / a.abs().max().clamp_min(1e-20)).item()
        worst = max(worst, e)
        # (bias gradients are atomically accumulated column sums: not bit-stable run to run either way; c_proj's bias gradient of the
        #  TOP layer of a call is the column sum of the bf16-rounded incoming gradient -- the cast in front of that layer's GEMMs --
        #  while below it comes unrounded out of the LayerNorm backward above: layer 2 heads the second chunk)
        assert e < (4e-3 if k.endswith('resblocks.2.mlp.c_proj.bias') else 5e-5), (k, e)
    print('worst relative difference of a parameter gradient, one call vs chunks of 3 layers:', worst)


def test_layernorm_backward_deferred_reduction():
    """mmvid_layernorm_bwd_partial + mmvid_layernorm_bwd_reduce_multi (what the tower backward does for its 2 x layers LayerNorms)
    against the single-call two-stage backward: dx identical, parameter / column-sum gradients bit-identical (same partial rows, same
    fixed-order reduction), null targets left alone, more entries than one launch holds."""
    from mmvid_amd import ops
    torch.manual_seed(2)
    rows, E, n = 3001, 768, 35
    items, want = [], []
    for i in range(n):
        x = torch.randn(rows, E, device=DEV)
        dy = torch.randn(rows, E, device=DEV)
        if i % 2:
            dy = dy.to(torch.bfloat16)
        w = torch.randn(E, device=DEV)
        mean, var = x.mean(1), x.var(1, unbiased=False)
        rstd = (var + 1e-5).rsqrt()
        g0 = torch.randn(rows, E, device=DEV)
        targets = [torch.randn(E, device=DEV) for _ in range(3)]
        use = [(i % 3) != 1, True, (i % 4) != 2]  # some entries without dw / colsum targets
        ref_t = [t.clone() for t in targets]
        ws_ref = torch.empty(512 * 3 * E, device=DEV)
        dx_ref = g0.clone()
        from mmvid_amd import _lib
        _lib.call('mmvid_layernorm_bwd_ex', ops._p(dy), int(dy.dtype == torch.bfloat16), E, ops._p(x), E, ops._p(mean), ops._p(rstd), ops._p(w),
                  rows, E, ops._p(dx_ref), E, 1, None, ops._p(ref_t[0]) if use[0] else None, ops._p(ref_t[1]), ops._p(ref_t[2]) if use[2] else None,
                  ops._p(ws_ref), ws_ref.numel(), ops._stream())
        ws = torch.empty(512 * 3 * E, device=DEV)
        dx = g0.clone()
        _, nb = ops.layernorm_bwd_partial(dy, x, mean, rstd, w, ws, dx=dx, add=True, want=use)
        assert torch.equal(dx, dx_ref)
        items.append((ws, targets[0] if use[0] else None, targets[1], targets[2] if use[2] else None))
        want.append((ref_t, targets, use))
    ops.layernorm_bwd_reduce_multi(items, nb, E)
    for ref_t, targets, use in want:
        for k in range(3):
            assert torch.equal(targets[k], ref_t[k]), 'deferred reduction differs from the single-call reduction'


@pytest.mark.parametrize('mode', ['split', 'mixed'])
@pytest.mark.parametrize('case', ['vqgan_tiny', 'vqgan_full', 'bert_tiny', 'bert_tiny_visual'])
def test_split_index_safety_margin(golden, case, mode):
    """How far the exact-index mode (`vae.strict = 'split'`) is from flipping an index, on every golden frame.  Only distance
    DIFFERENCES decide an argmin (|z|^2 is common to all codes), so per token: gap = d(z_ref, c2) - d(z_ref, c1) for the reference's
    best code c1 and its runner-up c2, err = |gap(z_split) - gap(z_ref)| (both in fp64 from the fp32 z: the encoder's error alone).
    The indices must be equal, and the gap must exceed SAFETY x err on every token; the histogram of gap / err is printed."""
    from mmvid_amd.vae import VQGanVAE1024
    from oracle import vqgan as ov
    from oracle.synth import synth_input
    from test_host_logic import tiny_vae
    SAFETY = 8.0
    g = golden(case)
    sets = []  # (vae module, frames [N,3,S,S], reference z [N*hw, C] or None (-> oracle), reference indices)
    if case.startswith('vqgan'):
        s = g.meta['image_size']
        vae = tiny_vae() if case == 'vqgan_tiny' else VQGanVAE1024(None, 128)
        vae.image_size = s
        load_synth(vae, g, 11)
        img = synth_input('img', (g.meta['n'], 3, s, s), 11, 'uniform')
        zr = g['z_e'].permute(0, 2, 3, 1).reshape(-1, g['z_e'].shape[1])
        sets.append((vae, img, zr, g['indices'].reshape(-1)))
    else:
        nv = 1 if case.endswith('visual') else 0
        m = load_synth(tiny_bert(nv, nv > 0), g, 17)
        for fr, tok in ((g['frames'], g['target_tok']), (g['warped_frames'], g['warp_tok'])):
            sets.append((m.vae, fr.reshape(-1, *fr.shape[2:]), None, tok.reshape(-1)))
        if nv:
            sets.append((m.cvae, g['visual'].reshape(-1, *g['visual'].shape[2:]), None, g['visual_tok'].reshape(-1)))
    ratios, worst = [], None
    for vae, img, zr, idx_ref in sets:
        e = vae.model.quantize.embedding.weight.detach().double().cpu()
        if zr is None:  # the oracle's fp32 encoder (pinned to the reference: tests/test_oracle_golden.py)
            sd = {'model.' + k: v.detach().float().cpu() for k, v in vae.model.state_dict().items()}
            z4 = ov.encode_z(sd, img, vae.image_size)
            zr = z4.permute(0, 2, 3, 1).reshape(-1, z4.shape[1])
        vae.strict = mode
        zs = vae.encode_z(img.to(DEV)).reshape(-1, zr.shape[1]).double().cpu()
        idx = vae.get_codebook_indices(img.to(DEV)).reshape(-1).cpu()
        vae.strict = False
        assert torch.equal(idx, idx_ref)
        zr = zr.double()

        def dist(z):  # without |z|^2 (common to every code)
            return (e * e).sum(1)[None, :] - 2.0 * z @ e.t()
        Dr, Ds = dist(zr), dist(zs)
        rows = torch.arange(zr.shape[0])
        d1 = Dr[rows, idx_ref]
        Dm = Dr.clone()
        Dm[rows, idx_ref] = float('inf')
        c2 = Dm.argmin(1)
        gap_r = Dr[rows, c2] - d1
        gap_s = Ds[rows, c2] - Ds[rows, idx_ref]
        err = (gap_s - gap_r).abs().clamp_min(1e-30)
        r = gap_r / err
        ratios.append(r)
        k = int(r.argmin())
        if worst is None or r[k] < worst[0]:
            worst = (r[k].item(), gap_r[k].item(), err[k].item())
    r = torch.cat(ratios)
    edges = [0, 1, 8, 64, 512, 4096, 1e30]
    hist = np.histogram(r.numpy(), edges)[0].tolist()
    print(f'{case}, {mode}: {r.numel()} tokens; reference top-2 gap / error of that gap: min {r.min().item():.1f} '
          f'(gap {worst[1]:.3e}, err {worst[2]:.3e}), median {r.median().item():.0f}; histogram (edges {edges[:-1]}): {hist}')
    assert r.min().item() > SAFETY


def test_lazy_rows_with_dense_table_exchange_and_skipped_step(golden):
    """(1) world > 1 with sparse_tables off -- the text-embedding gradient goes through the DENSE all-reduce, so rows
    only OTHER ranks touched carry a gradient here: the lazy-row flags must all be raised (Adam / grad-norm may skip nothing) and
    zero_grad() must clear the whole table.  World 2 is simulated on one device by a fake all-reduce that adds a peer's gradient.
    (2) a backward followed by zero_grad() WITHOUT step() must not leave its table rows behind."""
    from mmvid_amd.engine import FlatTrainer, backward_order
    g = golden('bert_tiny')
    text, frames = g['text'].to(DEV), g['frames'].to(DEV)

    def fwd_bwd(m):
        lm, lr, lv = m(text, target=frames, return_loss=True, rel=True, vid=True)
        (7 * lm + 0.5 * lr + 0.5 * lv).backward()

    # (1)
    m = load_synth(tiny_bert(), g, 17).train()
    tr = FlatTrainer(m, order=backward_order, sparse_tables=False)
    assert tr._lazy is not None
    W = m.text_emb.weight
    peer_rows = torch.tensor([3, 77, 40000], device=DEV)
    assert not bool(tr._lazy['flags'][peer_rows].any())
    tr.zero_grad()
    fwd_bwd(m)
    tr.world = 2

    def fake_send(lo, hi, tr=tr):  # the dense all-reduce of a 2-rank group: the peer's rows land in G
        if lo <= tr._lazy['lo'] < hi:
            W.grad[peer_rows] += 0.5
    tr._send, tr._exchange_sparse = fake_send, (lambda: None)
    before = W.detach()[peer_rows].clone()
    tr.step()
    assert bool(tr._lazy['flags'].all()), 'rows touched only by other ranks must count'
    assert not torch.equal(W.detach()[peer_rows], before), "Adam skipped a row that carried another rank's gradient"
    tr.zero_grad()
    assert float(W.grad.abs().sum()) == 0.0
    # (2)
    m2 = load_synth(tiny_bert(), g, 17).train()
    tr2 = FlatTrainer(m2, order=backward_order)
    tr2.zero_grad()
    fwd_bwd(m2)
    tr2.step()
    tr2.zero_grad()
    fwd_bwd(m2)  # e.g. a non-finite loss: the caller drops this step
    assert float(m2.text_emb.weight.grad.abs().sum()) > 0
    tr2.zero_grad()
    assert float(m2.text_emb.weight.grad.abs().sum()) == 0.0, 'zero_grad() after a skipped step left table rows behind'
    # (3) ADVICE r5: a backward whose forward ran BEFORE the previous zero_grad() (forward, zero_grad, backward, zero_grad): the id log
    # was reset in between, only the backward's own flag (BERT.table_grad_pending) knows that rows were written
    lm, lr, lv = m2(text, target=frames, return_loss=True, rel=True, vid=True)
    tr2.zero_grad()
    assert not m2.table_grad_pending
    (7 * lm + 0.5 * lr + 0.5 * lv).backward()
    assert m2.table_grad_pending and float(m2.text_emb.weight.grad.abs().sum()) > 0
    tr2.zero_grad()
    assert float(m2.text_emb.weight.grad.abs().sum()) == 0.0, 'rows of a backward whose forward preceded zero_grad() were left behind'
    # ... and two zero_grad() calls in a row decide "nothing to clear" without touching the table
    assert not m2.table_grad_pending
    tr2._lazy['all_dirty'] = False
    tr2.zero_grad()
    assert tr2._lazy['all_dirty'] is False
