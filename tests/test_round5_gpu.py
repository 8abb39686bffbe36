"""Round 5 on the GPU: the re-written attention kernels (mask-free tile loops, padding handled without the general predicate, tail
splits of all three passes), the GroupNorm finalisation folded into the apply launch, the software-pipelined LayerNorm backward, the
encoder's bf16 residual stream (index match rate against the reference before / after)."""
import numpy as np
import pytest
import torch

from test_models_gpu import DEV, close, load_synth

pytestmark = pytest.mark.gpu


def _attn_ref(qkv, B, L, H, mask):
    E = H * 64
    q, k, v = [t.view(B, L, H, 64).transpose(1, 2) for t in qkv.split(E, dim=1)]
    s = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s + mask
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, E)


def _mask_tensor(L, spec):
    if spec is None:
        return None
    if spec == 'causal':
        return torch.full((L, L), float('-inf'), device=DEV).triu_(1)
    m = torch.zeros(L, L, device=DEV)
    for r, c in spec[1]:
        m[r, :c] = float('-inf')
    return m


@pytest.mark.parametrize('pk', [1, 0])
@pytest.mark.parametrize('B,L,H,spec', [(1, 33, 2, None), (2, 97, 2, ('rows', [(70, 70), (71, 71)])), (2, 161, 3, None),
                                        (1, 577, 2, ('rows', [(65, 65), (66, 66)])), (1, 608, 2, None), (1, 609, 2, ('rows', [(129, 129), (130, 130)])),
                                        (1, 640, 2, None), (1, 641, 2, 'causal'), (2, 200, 2, 'causal'), (1, 643, 12, ('rows', [(129, 129), (130, 130)]))])
def test_attention_ragged_lengths_and_masks(B, L, H, spec, pk):
    """Sequence lengths on every side of the 32- / 64- / 128-position boundaries, all three mask shapes: the forward's padding compare,
    the backward passes that need no padding mask at all (zero-filled K rows in dQ, never-stored lanes in dK / dV), the mask-free tile
    pairs next to general tiles, both softmax instruction forms (option attn_pk) -- against fp32 torch on the bf16 inputs."""
    from mmvid_amd import _lib, ops
    E = H * 64
    torch.manual_seed(L + H)
    qkv = (torch.randn(B * L, 3 * E, device=DEV) * 0.7).bfloat16()
    dO = (torch.randn(B * L, E, device=DEV) * 0.2).bfloat16()
    qr = qkv.float().requires_grad_(True)
    ref = _attn_ref(qr, B, L, H, _mask_tensor(L, spec))
    ref.backward(dO.float())
    _lib.call('mmvid_set_option', b'attn_pk', pk)
    try:
        out, lse2 = ops.attention_fwd(qkv, B, L, H, spec)
        dqkv = ops.attention_bwd(qkv, out, dO, lse2, B, L, H, spec, workspace=False)
    finally:
        _lib.call('mmvid_set_option', b'attn_pk', 1)
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all() and torch.isfinite(lse2).all()
    close(out, ref, 1e-2, f'fwd L={L}')
    for nm, sl in (('dQ', slice(0, E)), ('dK', slice(E, 2 * E)), ('dV', slice(2 * E, 3 * E))):
        close(dqkv[:, sl], qr.grad[:, sl], 2e-2, f'{nm} L={L}')


def test_attention_forward_tail_split_matches_whole_blocks_and_is_reproducible():
    """mmvid_attention_fwd_ws at the training step's shape (1,080 blocks against 1,024 resident slots): the 56 blocks of the second
    round are cut into four key-range parts whose (O, max, sum) records are merged by a second launch.  Rows of those blocks differ from
    the workspace-free call by rounding only (one bf16 ulp), every other row is bit-identical, lse2 agrees to fp32 round-off, two runs
    are bit-identical; the backward of the split forward's (out, lse2) matches torch like the unsplit one."""
    from mmvid_amd import ops
    B, L, H, E = 18, 579, 12, 768
    torch.manual_seed(12)
    qkv = (torch.randn(B * L, 3 * E, device=DEV) * 0.5).bfloat16()
    spec = ('rows', [(65, 65), (66, 66)])
    whole, lse_w = ops.attention_fwd(qkv, B, L, H, spec)
    split, lse_s = ops.attention_fwd(qkv, B, L, H, spec, workspace=True)
    again, lse_a = ops.attention_fwd(qkv, B, L, H, spec, workspace=True)
    assert torch.equal(split, again) and torch.equal(lse_s, lse_a)
    w, s_ = whole.float(), split.float()
    diff = (w - s_).abs()
    frac = float((diff > 0).float().mean())
    print(f'forward tail split: {frac:.4f} of the outputs differ, max |d| {float(diff.max()):.3e}, max |d lse2| {float((lse_w - lse_s).abs().max()):.3e}')
    assert frac < 0.06  # 56 of 1,080 blocks
    assert bool((diff <= 2.0**-7 * w.abs() + 1e-5 * w.abs().max()).all())
    assert float((lse_w - lse_s).abs().max()) < 1e-4
    # a batch whose blocks fit the resident slots is never split: the call with a workspace IS the workspace-free call
    small = qkv[:4 * L].contiguous()
    a, la = ops.attention_fwd(small, 4, L, H, spec)
    b, lb = ops.attention_fwd(small, 4, L, H, spec, workspace=True)
    assert torch.equal(a, b) and torch.equal(la, lb)


@pytest.mark.parametrize('N,H,W,C,dt', [(3, 64, 64, 128, 'bf16'), (2, 32, 32, 256, 'f32'), (2, 128, 128, 128, 'bf16'), (5, 24, 24, 64, 'f32'),
                                        (2, 16, 16, 512, 'bf16')])
@pytest.mark.parametrize('swish', [True, False])
def test_groupnorm_finalisation_inside_the_apply_launch_is_bit_identical(N, H, W, C, dt, swish):
    """Option gn_fused (default 1): per-channel affine computed by every apply block from the partial sums (csrc/norm.hip
    groupnorm_apply_fused_kernel) against the separate finalize launch + one-chunk-per-thread apply -- the same operations in the same
    order: bit-identical outputs, own statistics pass and ragged pixel counts included."""
    from mmvid_amd import _lib, ops
    torch.manual_seed(C + H)
    x = torch.randn(N, H, W, C, device=DEV) * 1.5 + 0.3
    x = x.bfloat16() if dt == 'bf16' else x
    w, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    outs = []
    try:
        for flag in (0, 1):
            _lib.call('mmvid_set_option', b'gn_fused', flag)
            outs.append(ops.groupnorm_swish(x, w, b, swish=swish))
            outs.append(ops.groupnorm_swish(x, w, b, swish=swish, out_dtype=torch.float32))
    finally:
        _lib.call('mmvid_set_option', b'gn_fused', 1)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3])
    xf = x.float()
    g = xf.view(N, H * W, 32, C // 32)
    mu, var = g.mean((1, 3), keepdim=True), g.var((1, 3), unbiased=False, keepdim=True)
    ref = ((g - mu) * torch.rsqrt(var + 1e-6)).view(N, H, W, C) * w + b
    if swish:
        ref = ref * torch.sigmoid(ref)
    close(outs[3], ref, 2e-4, 'fused groupnorm vs torch')


def test_groupnorm_fold_with_conv_fused_statistics_keeps_the_encoder_bit_identical(golden):
    """The same switch through a whole encode (the partial sums then come from the convolutions' epilogues, per 128 / 64 pixels):
    z and the token indices are bit-identical."""
    from mmvid_amd import _lib
    from mmvid_amd.vae import VQGanVAE1024
    from oracle.synth import synth_input
    g = golden('vqgan_full')
    img = synth_input('img', (g.meta['n'], 3, 128, 128), 11, 'uniform').to(DEV)
    res = []
    try:
        for flag in (0, 1):
            _lib.call('mmvid_set_option', b'gn_fused', flag)
            vae = VQGanVAE1024(None, 128)
            vae.image_size = 128
            load_synth(vae, g, 11)
            res.append((vae.get_codebook_indices(img), vae.encode_z(img)))
    finally:
        _lib.call('mmvid_set_option', b'gn_fused', 1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize('E,rows', [(768, 10422), (768, 77), (512, 1000)])
@pytest.mark.parametrize('dy16', [True, False])
def test_layernorm_backward_pipelined_kernel_is_bit_identical(E, rows, dy16):
    """Option ln_fast (default 1): the software-pipelined LayerNorm backward (next row's operands requested before this row is reduced)
    against the generic kernel -- dx (added into the residual gradient), its bf16 copy and the partial rows of dw / db / colsum are
    bit-identical; few rows per wave, both dy precisions, both tower widths."""
    from mmvid_amd import _lib, ops
    torch.manual_seed(E + rows)
    x = torch.randn(rows, E, device=DEV) * 2 + 0.5
    w, b = torch.randn(E, device=DEV), torch.randn(E, device=DEV)
    _, mean, rstd = ops.layernorm_fwd(x, w, b)
    dy = torch.randn(rows, E, device=DEV)
    dy = dy.bfloat16() if dy16 else dy
    base = torch.randn(rows, E, device=DEV)
    res = []
    try:
        for flag in (0, 1):
            _lib.call('mmvid_set_option', b'ln_fast', flag)
            ws = torch.zeros(512 * 3 * E, device=DEV)
            dx, d16 = base.clone(), torch.zeros(rows, E, device=DEV, dtype=torch.bfloat16)
            _, nb = ops.layernorm_bwd_partial(dy, x, mean, rstd, w, ws, dx=dx, add=True, dx_bf16=d16)
            dw, db, cs = (torch.zeros(E, device=DEV) for _ in range(3))
            ops.layernorm_bwd_reduce_multi([(ws, dw, db, cs)], nb, E)
            # and the store form (no residual gradient to add to)
            dx2, _ = ops.layernorm_bwd_partial(dy, x, mean, rstd, w, torch.zeros(512 * 3 * E, device=DEV), want=(True, True, False))
            res.append((dx, d16, dw, db, cs, dx2))
    finally:
        _lib.call('mmvid_set_option', b'ln_fast', 1)
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (E, ), wr, b, 1e-5).backward(dy.float())
    close(res[1][0] - base, xr.grad, 1e-4, 'LN dx')
    close(res[1][2], wr.grad, 1e-4, 'LN dw')


def test_encoder_bf16_residual_stream_index_match_rate(golden):
    """vae.stream: 'bf16' (round 5, default) stores the encoder's residual stream between blocks as bf16, 'f32' keeps the fp32 stream of
    rounds 1-4; the exact modes are not touched.  Reports, against the reference's indices and z on the full-size golden frames, the
    match rate and z error of both (the before / after the round-4 review asks for) and holds both to the default mode's contract:
    every differing index is a near-tie of the reference's own top-2 distances."""
    from mmvid_amd.vae import VQGanVAE1024
    from oracle.synth import synth_input
    g = golden('vqgan_full')
    img = synth_input('img', (g.meta['n'], 3, 128, 128), 11, 'uniform').to(DEV)
    gap = (g['top2_d'][:, 1] - g['top2_d'][:, 0])
    rates = {}
    for stream in ('f32', 'bf16'):
        vae = VQGanVAE1024(None, 128)
        vae.image_size, vae.stream = 128, stream
        load_synth(vae, g, 11)
        idx = vae.get_codebook_indices(img).cpu()
        z = vae.encode_z(img).cpu()
        ref_z = g['z_e'].permute(0, 2, 3, 1)
        zerr = (z - ref_z).abs().max().item()
        zrel = ((z - ref_z).norm() / ref_z.norm()).item()
        mism = idx != g['indices']
        rates[stream] = 1.0 - float(mism.float().mean())
        print(f'encoder stream {stream}: {int(mism.sum())}/{idx.numel()} indices differ from the reference ({100 * rates[stream]:.2f} % match); '
              f'max |dz| {zerr:.3e}, relative L2 {zrel:.3e}')
        assert (gap.view_as(idx)[mism] < 64 * zerr + 1e-3).all()
        assert zrel < 3e-2
        vae.strict = 'split'  # the exact-index mode does not look at `stream`
        assert torch.equal(vae.get_codebook_indices(img).cpu(), g['indices'])
    assert rates['bf16'] > 0.9


def test_bert_golden_frames_token_match_with_both_streams(golden):
    """The same before / after on the frames of the BERT goldens (the 98.4 % of DESIGN.md section 4): tokens through the default
    encoder against the reference's, fp32 stream and bf16 stream."""
    from test_host_logic import tiny_bert
    for name, nv, cvae in (('bert_tiny', 0, False), ('bert_tiny_visual', 1, True)):
        g = golden(name)
        m = load_synth(tiny_bert(nv, cvae), g, 17).eval()
        frames = g['frames'].to(DEV)
        for stream in ('f32', 'bf16'):
            m.vae.stream = stream
            tt = m.get_image_tokens(frames).cpu()
            rate = float((tt == g['target_tok']).float().mean())
            print(f'{name}, encoder stream {stream}: {100 * rate:.2f} % of the reference tokens')
            assert rate > 0.9
