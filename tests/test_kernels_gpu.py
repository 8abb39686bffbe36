"""Per-kernel parity of the HIP path (through the C-ABI) on a real MI355X.

Bars: integer / index outputs bit-exact against the oracle; bf16-MFMA kernels within a relative
max-error of 2e-2 (bf16 operands, fp32 accumulate; stated per test) of an fp32 evaluation of the same
op on the same bf16-rounded inputs; fp32-in/fp32-out kernels within 1e-5..1e-4."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import relerr

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def ops():
    from mmvid_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def close(a, b, tol, what=''):
    e = relerr(a.float().cpu(), b.float().cpu())
    print(f'{what}: relerr {e:.3e} (tol {tol})')
    assert e <= tol, f'{what}: relerr {e} > {tol}'


# ------------------------------------------------------------------------------------------------ VQ
@pytest.mark.parametrize('tag,n', [('sep', 1024), ('stress', 1024), ('small', 256)])
def test_vq_argmin_bit_exact_vs_oracle(ops, golden, tag, n):
    from oracle.synth import synth_input, synth_tensor
    from oracle.vq import vq_argmin
    cb = (synth_input('cb_' + tag, (n, 256), 7, 'uniform') * 2 - 1) / n if tag == 'stress' else \
        synth_tensor('quantize.embedding.weight', (n, 256), 7)
    z = synth_input('z_' + tag, (512, 256), 7)
    idx_o, dmin_o = vq_argmin(z, cb)
    idx, dmin = ops.vq_argmin(z.to(DEV), cb.to(DEV), return_dmin=True)
    assert torch.equal(idx.cpu(), idx_o)
    assert torch.equal(dmin.cpu(), dmin_o)  # bit-exact distances, not just indices
    if tag != 'stress':
        assert torch.equal(idx.cpu(), golden('vq')[tag + '_idx'])  # and the reference itself


def test_vq_argmin_ragged_rows_and_ties(ops):
    from oracle.vq import vq_argmin
    g = torch.Generator().manual_seed(3)
    # row counts on both sides of the point where the codes are split over blocks as well (< 256 blocks of 32 rows): the
    # per-range minima are merged by atomicMin on (distance, index), which must reproduce the sequential first minimum
    for rows in (1, 31, 33, 3456, 4097, 8448):
        z = torch.randn(rows, 256, generator=g)
        cb = torch.randn(1024, 256, generator=g) * 0.5
        cb[700] = cb[5]  # exact duplicate rows: first index must win
        cb[900] = cb[5]
        idx_o, dmin_o = vq_argmin(z, cb)
        idx, dmin = ops.vq_argmin(z.to(DEV), cb.to(DEV), return_dmin=True)
        assert torch.equal(idx.cpu(), idx_o) and torch.equal(dmin.cpu(), dmin_o)
    z = cb[[5, 700, 900, 17]].clone()  # rows equal to a codebook entry
    assert ops.vq_argmin(z.to(DEV), cb.to(DEV)).cpu().tolist() == [5, 5, 5, 17]
    assert ops.vq_argmin(torch.empty(0, 256, device=DEV), cb.to(DEV)).numel() == 0
    bad = torch.randn(40, 256, generator=g)
    bad[7] = float('nan')  # no code beats +inf: index 0 and an infinite distance, with or without the split
    idx, dmin = ops.vq_argmin(bad.to(DEV), cb.to(DEV), return_dmin=True)
    assert int(idx[7]) == 0 and math.isinf(float(dmin[7])) and torch.equal(idx.cpu()[:7], vq_argmin(bad[:7], cb)[0])


def test_gather_rows(ops):
    t = rnd(1024, 256)
    idx = torch.randint(0, 1024, (7, 64), device=DEV)
    assert torch.equal(ops.gather_rows(t, idx), t[idx])
    assert torch.equal(ops.gather_rows(t, idx, torch.bfloat16), t[idx].bfloat16())


# ---------------------------------------------------------------------------------------------- GEMM
def _ref_mm(A, B, a_km, b_km):
    A32, B32 = A.float(), B.float()
    if a_km:
        A32 = A32.transpose(-1, -2)
    if not b_km:
        B32 = B32.transpose(-1, -2)
    return A32 @ B32


@pytest.mark.parametrize('a_km,b_km', [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (300, 136, 200), (1000, 768, 768), (579, 2304, 768)])
def test_gemm_layouts(ops, a_km, b_km, M, N, K):
    if a_km and M % 8:
        M = M // 8 * 8 + 8
    if (a_km or b_km) and K % 8:
        pass
    A = rnd(*((K, M) if a_km else (M, K)), seed=1, dtype=torch.bfloat16)
    B = rnd(*((K, N) if b_km else (N, K)), seed=2, dtype=torch.bfloat16)
    ref = _ref_mm(A, B, a_km, b_km)
    out = ops.gemm(A, B, a_kmajor=a_km, b_kmajor=b_km, out_dtype=torch.float32)
    close(out, ref, 1e-5 * math.sqrt(K) + 1e-4, f'gemm f32 {a_km}{b_km} {M}x{N}x{K}')
    out16 = ops.gemm(A, B, a_kmajor=a_km, b_kmajor=b_km)
    close(out16, ref, 1e-2, 'gemm bf16 out')


def test_gemm_k_reduction_not_multiple_of_tile_and_splitk(ops):
    # dW-shaped: reduce over 10422 tokens (not a multiple of 64), split-K with atomics into an existing buffer
    M, N, K = 776, 264, 1043
    A = rnd(K, M, seed=3, dtype=torch.bfloat16)
    B = rnd(K, N, seed=4, dtype=torch.bfloat16)
    base = rnd(M, N, seed=5)
    ref = base + _ref_mm(A, B, True, True)
    for sk in (1, 3, 8):
        out = base.clone()
        ops.gemm(A, B, a_kmajor=True, b_kmajor=True, out=out, accumulate=True, splitk=sk)
        close(out, ref, 2e-4, f'splitk={sk}')


def test_gemm_epilogues(ops):
    M, N, K = 333, 3072, 768
    A, W = rnd(M, K, seed=6, dtype=torch.bfloat16), rnd(N, K, seed=7, scale=0.05, dtype=torch.bfloat16)
    bias, res = rnd(N, seed=8), rnd(M, N, seed=9)
    pre = A.float() @ W.float().t() + bias
    # forward c_fc: save pre-activation, QuickGELU
    save = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    out = ops.gemm(A, W, bias=bias, act=1, save_pre=save)
    close(save, pre, 1e-2, 'save_pre')
    close(out, pre * torch.sigmoid(1.702 * pre), 1e-2, 'quickgelu')
    # residual + fp32 out
    out = ops.gemm(A, W, bias=bias, residual=res, out_dtype=torch.float32)
    close(out, pre + res, 1e-4, 'bias+residual f32')
    # backward through QuickGELU: dX-shaped NN gemm with dact
    dY = rnd(M, 768, seed=10, dtype=torch.bfloat16)
    Wp = rnd(768, N, seed=11, scale=0.05, dtype=torch.bfloat16)  # c_proj weight [E, F]
    x = save.float()
    s = torch.sigmoid(1.702 * x)
    ref = (dY.float() @ Wp.float()) * (s * (1 + 1.702 * x * (1 - s)))
    out = ops.gemm(dY, Wp, b_kmajor=True, dact_pre=save)
    close(out, ref, 1e-2, 'dgelu')
    # the same GEMM also delivering the column sums of its result (the bias gradient of the producing Linear)
    cs = rnd(N, seed=14)
    cs0 = cs.clone()
    out2 = ops.gemm(dY, Wp, b_kmajor=True, dact_pre=save, colsum=cs)
    assert torch.equal(out2, out)
    close(cs, cs0 + ref.sum(0), 2e-3, 'fused column sums')
    # batched
    Ab, Bb = rnd(3, 256, 64, seed=12, dtype=torch.bfloat16), rnd(3, 256, 64, seed=13, dtype=torch.bfloat16)
    close(ops.gemm(Ab, Bb, out_dtype=torch.float32), Ab.float() @ Bb.float().transpose(1, 2), 1e-4, 'batched')


def test_gemm_rejects_bad_shapes(ops):
    from mmvid_amd._lib import MMVIDError
    with pytest.raises(MMVIDError):
        ops.gemm(rnd(16, 12, dtype=torch.bfloat16), rnd(16, 12, dtype=torch.bfloat16))  # K % 8 != 0
    with pytest.raises(MMVIDError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))  # CPU tensors


# ---------------------------------------------------------------------------------------------- norms
def test_layernorm_fwd_bwd(ops):
    for rows, E in ((10, 768), (1158, 768), (77, 512)):
        x = rnd(rows, E, seed=1) * 2 + 0.5
        w, b = rnd(E, seed=2) * 0.1 + 1, rnd(E, seed=3) * 0.1
        y, mean, rstd = ops.layernorm_fwd(x, w, b, out_dtype=torch.float32)
        xr = x.clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.layer_norm(xr, (E, ), wr, br, 1e-5)
        close(y, yr, 1e-5, 'ln fwd')
        y16, _, _ = ops.layernorm_fwd(x, w, b)
        assert torch.equal(y16, y.bfloat16())
        dy = rnd(rows, E, seed=4)
        yr.backward(dy)
        base = rnd(rows, E, seed=5)
        dx = base.clone()
        dw, db = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
        cs = rnd(E, seed=6)
        cs0 = cs.clone()
        ops.layernorm_bwd(dy, x, mean, rstd, w, dx=dx, add=True, dw=dw, db=db, dx_colsum=cs)
        close(dx, base + xr.grad, 1e-5, 'ln dx')
        close(cs, cs0 + dx.sum(0), 1e-5, 'ln colsum(dx)')
        close(dw, wr.grad, 1e-4, 'ln dw')
        close(db, br.grad, 1e-4, 'ln db')
        # two-stage (workspace) reduction: same values, deterministic (bit-identical between runs), partial outputs allowed
        ws = torch.empty(3 * E * 256, device=DEV)
        outs = []
        for _ in range(2):
            dx2, dw2, db2, cs2 = base.clone(), torch.zeros(E, device=DEV), torch.zeros(E, device=DEV), cs0.clone()
            ops.layernorm_bwd(dy, x, mean, rstd, w, dx=dx2, add=True, dw=dw2, db=db2, dx_colsum=cs2, workspace=ws)
            outs.append((dx2, dw2, db2, cs2))
        assert all(torch.equal(a, b_) for a, b_ in zip(*outs))
        assert torch.equal(outs[0][0], dx)
        close(outs[0][1], wr.grad, 1e-5, 'ln dw (two-stage)'), close(outs[0][2], br.grad, 1e-5, 'ln db (two-stage)')
        close(outs[0][3], cs0 + dx.sum(0), 1e-5, 'ln colsum (two-stage)')
        cs3 = cs0.clone()
        ops.layernorm_bwd(dy, x, mean, rstd, w, dx=base.clone(), add=True, dx_colsum=cs3, workspace=ws)
        assert torch.equal(cs3, outs[0][3])


def test_groupnorm_swish(ops):
    for N, H, C in ((2, 16, 128), (1, 8, 512), (3, 4, 32), (1, 32, 256)):
        x = rnd(N, H, H, C, seed=1) * 1.5 + 0.3
        w, b = rnd(C, seed=2) * 0.1 + 1, rnd(C, seed=3) * 0.1
        ref = F.group_norm(x.permute(0, 3, 1, 2), 32, w, b, 1e-6)
        ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1)
        close(ops.groupnorm_swish(x, w, b, out_dtype=torch.float32), ref, 2e-5, f'gn f32 C={C}')
        x16 = x.bfloat16()
        ref16 = F.group_norm(x16.float().permute(0, 3, 1, 2), 32, w, b, 1e-6).permute(0, 2, 3, 1)
        close(ops.groupnorm_swish(x16, w, b, swish=False), ref16, 1e-2, 'gn bf16 noswish')


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, B, L, H, mask):
    E = H * 64
    q, k, v = qkv.float().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * L, E), p


def _mask_tensor(L, spec):
    if spec is None:
        return None
    if spec == 'causal':
        return torch.full((L, L), float('-inf'), device=DEV).triu_(1)
    m = torch.zeros(L, L, device=DEV)
    for r, c in spec[1]:
        m[r, :c] = float('-inf')
    return m


@pytest.mark.parametrize('B,L,H,spec', [(2, 51, 12, ('rows', [(17, 17), (18, 18)])), (2, 579, 12, ('rows', [(65, 65), (66, 66)])),
                                        (1, 130, 8, 'causal'), (3, 64, 2, None), (1, 1152, 12, 'causal')])
def test_attention_fwd_bwd(ops, B, L, H, spec):
    E = H * 64
    qkv = rnd(B * L, 3 * E, seed=L, dtype=torch.bfloat16)
    qkv_r = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(qkv_r, B, L, H, _mask_tensor(L, spec))
    out, lse2 = ops.attention_fwd(qkv, B, L, H, spec)
    close(out, ref, 1e-2, f'attn fwd L={L}')
    dout = rnd(B * L, E, seed=L + 1, dtype=torch.bfloat16)
    ref.backward(dout.float())
    dqkv = ops.attention_bwd(qkv, out, dout, lse2, B, L, H, spec)
    g = qkv_r.grad
    close(dqkv[:, :E], g[:, :E], 2e-2, 'dQ')
    close(dqkv[:, E:2 * E], g[:, E:2 * E], 2e-2, 'dK')
    close(dqkv[:, 2 * E:], g[:, 2 * E:], 2e-2, 'dV')


@pytest.mark.parametrize('spike_at', [40, 150, 190])
def test_attention_softmax_spike(ops, spike_at):
    # one key dominates late in the sequence: forces the (lazy, thresholded) online-softmax rescale branch to be
    # taken at a chosen tile, with a jump far beyond the 2^8 threshold for some rows and below it for others
    B, L, H = 1, 200, 1
    qkv = rnd(B * L, 192, seed=9, dtype=torch.bfloat16)
    q = qkv[:, :64].float()
    qkv[spike_at, 64:128] = (q[3] * 6).bfloat16()
    qkv[spike_at - 7, 64:128] = (q[100] * 0.8).bfloat16()  # a mild bump: stays under the threshold
    qkv_r = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(qkv_r, B, L, H, None)
    out, lse2 = ops.attention_fwd(qkv, B, L, H, None)
    close(out, ref, 1e-2, 'spike fwd')
    # lse2 is the log2-domain logsumexp whatever reference maximum the kernel kept
    s = (qkv[:, :64].float() @ qkv[:, 64:128].float().t()) * 0.125
    close(lse2.view(-1), torch.logsumexp(s, -1) * 1.4426950408889634, 1e-4, 'spike lse2')
    dout = rnd(B * L, 64, seed=10, dtype=torch.bfloat16)
    ref.backward(dout.float())
    dqkv = ops.attention_bwd(qkv, out, dout, lse2, B, L, H, None)
    close(dqkv, qkv_r.grad, 2e-2, 'spike bwd')


def test_attention_deterministic(ops):
    B, L, H = 3, 579, 12
    qkv = rnd(B * L, 3 * H * 64, seed=77, dtype=torch.bfloat16)
    dout = rnd(B * L, H * 64, seed=78, dtype=torch.bfloat16)
    spec = ('rows', [(65, 65), (66, 66)])
    outs = []
    for _ in range(4):
        out, lse2 = ops.attention_fwd(qkv, B, L, H, spec)
        outs.append((out, lse2, ops.attention_bwd(qkv, out, dout, lse2, B, L, H, spec)))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))


def test_attention_reproducible_in_graph_replay_under_memory_pressure(ops):
    """Regression test of an LDS-DMA race: attn_fwd_kernel's tile loop ran without an s_waitcnt vmcnt before its barrier, so
    a K/V piece that landed late was read as stale LDS -- never seen with eager launches, once in a few hundred launches when
    the step was replayed as a hipGraph (kernels back to back, copies in flight): NaN rows in one layer, every parameter NaN
    after the clip.  The kernels have no atomics, so every replay must reproduce the first result bit for bit."""
    B, L, H = 18, 579, 12  # BASELINE config 2: three passes of six sequences
    qkv = rnd(B * L, 3 * H * 64, seed=91, dtype=torch.bfloat16)
    dout = rnd(B * L, H * 64, seed=92, dtype=torch.bfloat16)
    spec = ('rows', [(65, 65), (66, 66)])
    big_a, big_b = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)
    ref_out, ref_lse = ops.attention_fwd(qkv, B, L, H, spec)
    ref_dqkv = ops.attention_bwd(qkv, ref_out, dout, ref_lse, B, L, H, spec)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        res = []
        with torch.cuda.graph(g, stream=side):
            for _ in range(12):
                big_b.copy_(big_a)  # keeps the memory system busy right in front of the attention kernels
                out, lse2 = ops.attention_fwd(qkv, B, L, H, spec)
                res.append((out, lse2, ops.attention_bwd(qkv, out, dout, lse2, B, L, H, spec)))
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(40):
        g.replay()
        torch.cuda.synchronize()
        for i, (out, lse2, dqkv) in enumerate(res):
            assert torch.equal(out, ref_out) and torch.equal(lse2, ref_lse) and torch.equal(dqkv, ref_dqkv), (rep, i)


def test_mfma_kernels_reproducible_in_graph_replay_under_memory_pressure(ops):
    """The same hazard check as the attention test, for every other kernel that stages tiles by LDS-DMA: the three GEMM
    layouts (dW with the deterministic split-K), the implicit-GEMM convolution (one-pass and split-K) and the strip
    convolution, replayed back to back as a hipGraph with copy traffic in between.  None of them uses atomics on these
    paths, so every replay must equal the first result bit for bit."""
    M = 10422
    X = rnd(M, 768, seed=1, dtype=torch.bfloat16)
    W = rnd(2304, 768, seed=2, scale=0.03, dtype=torch.bfloat16)
    dY = rnd(M, 2304, seed=3, dtype=torch.bfloat16)
    bias = rnd(2304, seed=4)
    x8 = rnd(54, 8, 8, 512, seed=5, dtype=torch.bfloat16)
    w8 = rnd(512, 9, 512, seed=6, scale=0.02, dtype=torch.bfloat16)
    x32 = rnd(8, 32, 32, 256, seed=7, dtype=torch.bfloat16)
    w32 = rnd(256, 9, 256, seed=8, scale=0.02, dtype=torch.bfloat16)
    b512, b256 = rnd(512, seed=9), rnd(256, seed=10)
    big_a, big_b = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)

    def work():
        dW = torch.zeros(2304, 768, device=DEV)
        return (ops.gemm(X, W, bias=bias), ops.gemm(dY, W, b_kmajor=True, out_dtype=torch.float32), ops.gemm_dw(dY, X, dW),
                ops.conv2d_nhwc(x8, w8, b512, 0, out_dtype=torch.float32), ops.conv2d_nhwc(x8, w8, b512, 0, splitk=4),
                ops.conv2d_nhwc(x32, w32, b256, 0), ops.conv3x3_strip(x32, w32, b256))

    ref = [t.clone() for t in work()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            res = []
            for _ in range(6):
                big_b.copy_(big_a)
                res.append(work())
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(25):
        g.replay()
        torch.cuda.synchronize()
        for i, outs in enumerate(res):
            for k, (a, b) in enumerate(zip(outs, ref)):
                assert torch.equal(a, b), (rep, i, k)


# ------------------------------------------------------------------------------------- embed / losses
def test_out_of_range_ids_are_counted_not_silent(ops):
    """An embedding id outside its table reads row 0 and a CE target outside [0, V) counts as class 0 (a kernel must never
    fault), but both are counted on the device and _lib.check_device_faults() raises (the reference hits a device assert)."""
    from mmvid_amd import _lib
    _lib.device_faults(reset=True)
    E = 64
    tabs = [rnd(5, E, seed=1), rnd(7, E, seed=2)]
    seg = torch.tensor([0, 1, 1], dtype=torch.int32, device=DEV)
    ids = torch.tensor([[1, 6, 2], [4, 7, -1]], device=DEV)  # 7 and -1 are outside table 1 (7 rows)
    x = ops.assemble_sequence(tabs, ids, seg, torch.zeros(3, E, device=DEV))
    assert torch.equal(x[1, 1], tabs[1][0]) and torch.equal(x[1, 2], tabs[1][0]) and torch.equal(x[0, 1], tabs[1][6])
    logits = rnd(4, 16, seed=3)
    ops.cross_entropy_fwd(logits, torch.tensor([0, 15, 16, 3], device=DEV), None)
    assert _lib.device_faults(reset=False)[:2] == [2, 1]
    with pytest.raises(_lib.MMVIDError, match='out-of-range'):
        _lib.check_device_faults()
    assert _lib.device_faults() == [0, 0, 0, 0]  # the check cleared them


def test_assemble_sequence_and_backward(ops):
    B, L, E = 3, 37, 768
    tabs = [rnd(5, E, seed=1), rnd(200, E, seed=2), rnd(258, E, seed=3)]
    seg = torch.tensor([0] + [1] * 10 + [0, 0] + [2] * 24, dtype=torch.int32, device=DEV)
    ids = torch.stack([torch.randint(0, tabs[s].shape[0], (B, ), device=DEV) for s in seg.tolist()], 1)
    pos = rnd(L, E, seed=4)
    x = ops.assemble_sequence(tabs, ids, seg, pos)
    ref = torch.stack([torch.stack([tabs[seg[l]][ids[b, l]] + pos[l] for l in range(L)]) for b in range(B)])
    assert torch.equal(x, ref)
    dx = rnd(B, L, E, seed=5)
    gts = [torch.zeros_like(t) for t in tabs]
    dpos = torch.empty(L, E, device=DEV)
    ops.assemble_sequence_bwd(gts, [t.shape[0] for t in tabs], ids, seg, dx, dpos)
    close(dpos, dx.sum(0), 1e-6, 'dpos')
    for s, gt in enumerate(gts):
        r = torch.zeros_like(gt)
        for l in range(L):
            if seg[l] == s:
                r.index_add_(0, ids[:, l], dx[:, l])
        close(gt, r, 1e-6, f'dtable{s}')
    # heavy duplication (every target position holds the same id, as [MASK] does) and a table without gradient
    B = 40
    ids = torch.stack([torch.randint(0, tabs[s].shape[0], (B, ), device=DEV) for s in seg.tolist()], 1)
    ids[:, seg == 2] = 7
    dx = rnd(B, L, E, seed=6)
    gts = [torch.zeros_like(tabs[0]), None, torch.zeros_like(tabs[2])]
    ops.assemble_sequence_bwd(gts, [t.shape[0] for t in tabs], ids, seg, dx, None)
    r = torch.zeros_like(tabs[2])
    r[7] = dx[:, seg == 2].sum((0, 1))
    close(gts[2], r, 1e-5, 'dtable2, one hot row')
    r0 = torch.zeros_like(tabs[0])
    for l in range(L):
        if seg[l] == 0:
            r0.index_add_(0, ids[:, l], dx[:, l])
    close(gts[0], r0, 1e-5, 'dtable0 next to a frozen table')


def test_cross_entropy(ops):
    rows, V = 1000, 1024
    logits = rnd(rows, V, seed=1) * 3
    target = torch.randint(0, V, (rows, ), device=DEV)
    sel = (torch.rand(rows, device=DEV) < 0.7)
    lr = logits.clone().requires_grad_(True)
    loss_ref = F.cross_entropy(lr[sel], target[sel])
    lse, loss_sum = ops.cross_entropy_fwd(logits, target, sel.to(torch.uint8))
    cnt = sel.sum()
    close(loss_sum / cnt, loss_ref.detach().view(1), 1e-5, 'ce loss')
    loss_ref.backward()
    gs = (1.0 / cnt).float().view(1)
    d = ops.cross_entropy_bwd(logits, target, sel.to(torch.uint8), lse, gs)
    close(d, lr.grad, 1e-2, 'ce grad')
    assert (d[~sel] == 0).all()


def test_adam_and_grad_norm(ops):
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2) * 3
    p = torch.cat([p, torch.zeros(1, device=DEV)])[:n].contiguous()
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    for step in (1, 2, 3):
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        sq = torch.zeros(1, device=DEV)
        ops.grad_sqnorm(g, sq)
        close(sq, (g.double()**2).sum().float().view(1), 1e-5, 'sqnorm')
        ops.adam_step(p, g, m, v, shadow, step, 1e-3, max_norm=1.0, sqnorm=sq)
        close(p, pr.detach(), 1e-6, f'adam step {step}')
        assert torch.equal(shadow, p.bfloat16())


# ---------------------------------------------------------------------------------------------- VQGAN
@pytest.mark.parametrize('mode', [0, 1, 2, 3])
@pytest.mark.parametrize('N,H,Cin,Cout', [(2, 16, 128, 128), (1, 8, 256, 512), (3, 4, 32, 64), (1, 32, 8, 128), (1, 16, 128, 8)])
def test_conv_modes(ops, mode, N, H, Cin, Cout):
    taps = 1 if mode == 3 else 9
    x = rnd(N, H, H, Cin, seed=1, dtype=torch.bfloat16)
    w = rnd(Cout, taps, Cin, seed=2, scale=1 / math.sqrt(taps * Cin), dtype=torch.bfloat16)
    b = rnd(Cout, seed=3) * 0.1
    xn = x.float().permute(0, 3, 1, 2)
    wn = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2) if mode != 3 else w.float().view(Cout, Cin, 1, 1)
    if mode == 0:
        ref = F.conv2d(xn, wn, b, padding=1)
    elif mode == 1:
        ref = F.conv2d(F.pad(xn, (0, 1, 0, 1)), wn, b, stride=2)
    elif mode == 2:
        ref = F.conv2d(F.interpolate(xn, scale_factor=2.0, mode='nearest'), wn, b, padding=1)
    else:
        ref = F.conv2d(xn, wn, b)
    ref = ref.permute(0, 2, 3, 1)
    out = ops.conv2d_nhwc(x, w, b, mode, out_dtype=torch.float32)
    close(out, ref, 2e-4, f'conv mode {mode}')
    res = rnd(*ref.shape, seed=4, dtype=torch.bfloat16)
    out = ops.conv2d_nhwc(x, w, b, mode, residual=res)
    close(out, ref + res.float(), 1e-2, 'conv + residual bf16')
    out = ops.conv2d_nhwc(x, w, b, mode, clamp01=True, out_dtype=torch.float32)
    close(out, (ref.clamp(-1, 1) + 1) * 0.5, 2e-4, 'conv clamp01')


@pytest.mark.parametrize('mode,N,H,Cin,Cout', [(0, 54, 8, 512, 512), (0, 3, 8, 256, 512), (0, 5, 8, 64, 264), (1, 2, 16, 128, 128),
                                               (3, 7, 8, 512, 256), (0, 1, 8, 8, 8)])
def test_conv_split_k(ops, mode, N, H, Cin, Cout):
    """Split-K convolution (deep layers on 8x8 maps): same values as the one-pass kernel up to fp32 summation order, every
    epilogue (bias, bf16 / fp32 residual, clamp, both output precisions), ragged row counts, splits that get no K tile
    (K = 72: two 64-deep tiles for four splits), and bit-reproducible between launches."""
    taps = 1 if mode == 3 else 9
    x = rnd(N, H, H, Cin, seed=1, dtype=torch.bfloat16)
    w = rnd(Cout, taps, Cin, seed=2, scale=1 / math.sqrt(taps * Cin), dtype=torch.bfloat16)
    b = rnd(Cout, seed=3) * 0.1
    one = ops.conv2d_nhwc(x, w, b, mode, out_dtype=torch.float32)
    res16, res32 = rnd(*one.shape, seed=4, dtype=torch.bfloat16), rnd(*one.shape, seed=5)
    for sk in (2, 4):
        out = ops.conv2d_nhwc(x, w, b, mode, out_dtype=torch.float32, splitk=sk)
        close(out, one, 2e-5, f'split-K {sk} vs one pass')
        assert torch.equal(out, ops.conv2d_nhwc(x, w, b, mode, out_dtype=torch.float32, splitk=sk))
        close(ops.conv2d_nhwc(x, w, b, mode, residual=res16, splitk=sk), ops.conv2d_nhwc(x, w, b, mode, residual=res16), 1e-2,
              'split-K + bf16 residual, bf16 out')
        close(ops.conv2d_nhwc(x, w, b, mode, residual=res32, out_dtype=torch.float32, splitk=sk), one + res32, 2e-5,
              'split-K + fp32 residual')
        close(ops.conv2d_nhwc(x, w, b, mode, clamp01=True, out_dtype=torch.float32, splitk=sk), (one.clamp(-1, 1) + 1) * 0.5, 2e-5,
              'split-K clamp01')


@pytest.mark.parametrize('mode,N,H,Cin,Cout,out32', [(0, 3, 16, 128, 128, True), (0, 2, 32, 64, 256, False),
                                                     (1, 5, 32, 128, 128, True), (3, 2, 16, 256, 512, False),
                                                     (0, 1, 48, 128, 128, False)])
def test_conv_fused_groupnorm_statistics(ops, mode, N, H, Cin, Cout, out32):
    """The conv epilogue's GroupNorm partial sums (ragged tiles: 5*256 and 48*48 pixels are not multiples of 256)
    give the same normalisation as the standalone statistics pass, and both match torch."""
    taps = 1 if mode == 3 else 9
    x = rnd(N, H, H, Cin, seed=1, dtype=torch.bfloat16)
    w = rnd(Cout, taps, Cin, seed=2, scale=1 / math.sqrt(taps * Cin), dtype=torch.bfloat16)
    b = rnd(Cout, seed=3) * 0.1
    gw, gb = rnd(Cout, seed=4) * 0.1 + 1, rnd(Cout, seed=5) * 0.1
    Ho = H // 2 if mode == 1 else H
    dt = torch.float32 if out32 else torch.bfloat16
    res = rnd(N, Ho, Ho, Cout, seed=6, dtype=dt)
    st = ops.gn_stats_buffer(N, Ho * Ho, Cout, x.device)
    y = ops.conv2d_nhwc(x, w, b, mode, residual=res, out_dtype=dt, gn_stats=st)
    assert torch.equal(y, ops.conv2d_nhwc(x, w, b, mode, residual=res, out_dtype=dt))  # the output itself is unchanged
    fused = ops.groupnorm_swish(y, gw, gb, out_dtype=torch.float32, stats=st)
    plain = ops.groupnorm_swish(y, gw, gb, out_dtype=torch.float32)
    ref = F.group_norm(y.float().permute(0, 3, 1, 2), 32, gw, gb, 1e-6)
    ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1)
    close(fused, ref, 2e-5, 'gn(fused stats) vs torch')
    close(fused, plain, 1e-5, 'gn(fused stats) vs gn(own stats)')
    st2 = ops.gn_stats_buffer(N, Ho * Ho, Cout, x.device)
    ops.conv2d_nhwc(x, w, b, mode, residual=res, out_dtype=dt, gn_stats=st2)
    off, used = N * Cout * 2, N * ((Ho * Ho + 127) // 128) * 64  # the buffer is sized for 64-pixel blocks; 128-pixel producers fill half
    assert torch.equal(st[off:off + used], st2[off:off + used])  # deterministic partial sums


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 32, 32, 128, 128), (1, 64, 64, 128, 128), (3, 32, 32, 256, 256), (1, 128, 128, 128, 128),
                                            (2, 32, 32, 128, 256), (5, 16, 16, 256, 256), (3, 16, 16, 64, 128), (7, 8, 8, 512, 512),
                                            (1, 64, 32, 32, 128)])
def test_conv3x3_strip_vs_torch(ops, N, H, W, Cin, Cout):
    """csrc/conv_strip.hip: the (kernel row, 32 channels) K tiling with the input strip staged once per three taps.  Covers
    every strip geometry (image rows per 512-pixel tile: 4 ... 64, tiles spanning several images, ragged last tile),
    zero padding at all four borders, both residual precisions, dual stores and the 64-pixel GroupNorm partial sums."""
    x = rnd(N, H, W, Cin, seed=1, dtype=torch.bfloat16)
    w = rnd(Cout, 9, Cin, seed=2, scale=1 / math.sqrt(9 * Cin), dtype=torch.bfloat16)
    b = rnd(Cout, seed=3) * 0.1
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    out = ops.conv3x3_strip(x, w, b, out_dtype=torch.float32)
    close(out, ref, 2e-4, f'strip conv {H}x{W} {Cin}->{Cout}')
    old = ops.conv2d_nhwc(x, w, b, 0, out_dtype=torch.float32)
    close(out, old, 1e-5, 'strip vs per-tap kernel (same bf16 products, different fp32 summation order)')
    r32, r16 = rnd(N, H, W, Cout, seed=4), rnd(N, H, W, Cout, seed=5, dtype=torch.bfloat16)
    st = ops.gn_stats_buffer(N, H * W, Cout, x.device)
    o32, o16 = ops.conv3x3_strip(x, w, b, residual=r32, out_dtype=torch.float32, also_bf16=True, gn_stats=st)
    close(o32, ref + r32, 2e-4, 'strip + fp32 residual')
    assert torch.equal(o16, o32.bfloat16())
    gw, gb = rnd(Cout, seed=6) * 0.1 + 1, rnd(Cout, seed=7) * 0.1
    fused = ops.groupnorm_swish(o32, gw, gb, out_dtype=torch.float32, stats=st, stats_block=64)
    close(fused, ops.groupnorm_swish(o32, gw, gb, out_dtype=torch.float32), 1e-5, 'gn(strip partial sums) vs gn(own stats)')
    st16 = ops.gn_stats_buffer(N, H * W, Cout, x.device)
    y16 = ops.conv3x3_strip(x, w, b, residual=r16, gn_stats=st16)
    close(y16, ref + r16.float(), 1e-2, 'strip + bf16 residual, bf16 out')
    close(ops.groupnorm_swish(y16, gw, gb, out_dtype=torch.float32, stats=st16, stats_block=64),
          ops.groupnorm_swish(y16, gw, gb, out_dtype=torch.float32), 1e-5, 'gn on the bf16 output')
    # batch independence and determinism
    assert torch.equal(ops.conv3x3_strip(x[:1].contiguous(), w, b, out_dtype=torch.float32), out[:1])
    assert torch.equal(ops.conv3x3_strip(x, w, b, out_dtype=torch.float32), out)


def test_image_layout_kernels(ops):
    img = torch.rand(2, 3, 16, 16, device=DEV)
    o = ops.image_to_nhwc8(img)
    assert torch.equal(o[..., :3], (2 * img - 1).permute(0, 2, 3, 1).bfloat16()) and (o[..., 3:] == 0).all()
    x = rnd(2, 8, 8, 8)
    assert torch.equal(ops.nhwc_to_nchw(x, 3), x.permute(0, 3, 1, 2)[:, :3].contiguous())


@pytest.mark.parametrize('N,HW,C', [(2, 256, 256), (1, 64, 512), (3, 16, 128)])
def test_spatial_attention(ops, N, HW, C):
    q, k, v = (rnd(N, HW, C, seed=s, dtype=torch.bfloat16) for s in (1, 2, 3))
    p = torch.softmax(q.float() @ k.float().transpose(1, 2) * C**-0.5, -1)
    close(ops.spatial_attention(q, k, v), p @ v.float(), 1.5e-2, 'spatial attn')


def test_colsum_and_dw_workspace(ops):
    for M, N in ((10422, 768), (300, 3072), (1, 8), (257, 264)):
        dy = rnd(M, N, seed=M, dtype=torch.bfloat16)
        db = rnd(N, seed=1)
        ref = db + dy.float().sum(0)
        ops.colsum_bf16(dy, db)
        close(db, ref, 2e-5, f'colsum {M}x{N}')
    # deterministic split-K dW: bit-identical across runs and across split factors' own repetitions
    dY, X = rnd(3000, 768, seed=3, dtype=torch.bfloat16), rnd(3000, 256, seed=4, dtype=torch.bfloat16)
    ref = dY.float().t() @ X.float()
    outs = []
    for _ in range(3):
        dW = torch.zeros(768, 256, device=DEV)
        ops.gemm_dw(dY, X, dW, accumulate=True, splitk=5)
        outs.append(dW)
    close(outs[0], ref, 2e-4, 'dW splitk=5')
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    base = rnd(768, 256, seed=5)
    dW = base.clone()
    ops.gemm_dw(dY, X, dW, accumulate=True)
    close(dW, base + ref, 2e-4, 'dW accumulate auto-split')
