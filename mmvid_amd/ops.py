"""Tensor-level wrappers over the C-ABI (mmvid_amd/_lib.py).  PyTorch here is plumbing only: it owns the
device memory and the stream; every computation below runs in the hand-written HIP kernels.

All tensors must live on a HIP device and be contiguous (strided views are passed through explicit ld
arguments where a kernel supports them)."""
import ctypes

import torch

from . import _lib
from ._lib import call

bf16, f32, i64 = torch.bfloat16, torch.float32, torch.int64


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise _lib.MMVIDError(f'{name}: tensor is on {t.device}; the MMVID kernels run on an MI355X only (no CPU path)')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: tensor must be contiguous')
    return t


# ------------------------------------------------------------------------------------------------ VQ
def vq_sqnorm(codebook):
    _chk(codebook, f32, 'codebook')
    ee = torch.empty(codebook.shape[0], device=codebook.device, dtype=f32)
    call('mmvid_vq_sqnorm', _p(codebook), codebook.shape[0], codebook.shape[1], _p(ee), _stream())
    return ee


def vq_argmin(z, codebook, ee=None, return_dmin=False):
    """z [rows, 256] f32, codebook [n, 256] f32 -> idx int64 [rows] (first argmin of the reference expression)."""
    _chk(z, f32, 'z'), _chk(codebook, f32, 'codebook')
    if ee is None:
        ee = vq_sqnorm(codebook)
    rows = z.shape[0]
    idx = torch.empty(rows, device=z.device, dtype=i64)
    dmin = torch.empty(rows, device=z.device, dtype=f32) if return_dmin else None
    if rows == 0:
        return (idx, dmin) if return_dmin else idx
    call('mmvid_vq_argmin_l2', _p(z), _p(codebook), _p(ee), rows, codebook.shape[0], codebook.shape[1], _p(idx),
         _p(dmin), _stream())
    return (idx, dmin) if return_dmin else idx


def gather_rows(table, idx, out_dtype=f32):
    _chk(table, f32, 'table'), _chk(idx, i64, 'idx')
    rows, dim = idx.numel(), table.shape[1]
    out = torch.empty(*idx.shape, dim, device=table.device, dtype=out_dtype)
    call('mmvid_gather_rows', _p(table), table.shape[0], _p(idx), rows, dim, _p(out) if out_dtype == f32 else None,
         _p(out) if out_dtype == bf16 else None, _stream())
    return out


# ---------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, *, a_kmajor=False, b_kmajor=False, bias=None, residual=None, dact_pre=None, save_pre=None, act=0,
         out_dtype=bf16, out=None, accumulate=False, splitk=1, alpha=1.0, colsum=None):
    """C[m,n] = sum_k A(m,k) B(n,k).  A: [M,K] (or [K,M] when a_kmajor), B: [N,K] (or [K,N] when b_kmajor), 2-D
    bf16; or 3-D batched with identical leading batch size."""
    _chk(A, bf16, 'A'), _chk(B, bf16, 'B')
    batch = 1
    sA = sB = 0
    if A.dim() == 3:
        batch = A.shape[0]
        sA, sB = A.stride(0), B.stride(0)
        A2, B2 = A[0], B[0]
    else:
        A2, B2 = A, B
    K, M = (A2.shape if a_kmajor else A2.shape[::-1])
    Kb, N = (B2.shape if b_kmajor else B2.shape[::-1])
    assert K == Kb, (A.shape, B.shape)
    if out is None:
        shape = (batch, M, N) if A.dim() == 3 else (M, N)
        out = torch.empty(shape, device=A.device, dtype=out_dtype)
    sC = M * N if batch > 1 else 0
    is32 = out.dtype == f32
    for t, n in ((bias, 'bias'), (residual, 'residual')):
        if t is not None:
            _chk(t, f32, n)
    call('mmvid_gemm_bf16', int(a_kmajor), int(b_kmajor), M, N, K, _p(A), A2.stride(0), _p(B), B2.stride(0), batch,
         sA, sB, sC, splitk, float(alpha), _p(bias), _p(residual), N, _p(dact_pre), _p(save_pre), N, act,
         int(accumulate), _p(out) if is32 else None, None if is32 else _p(out), N, _p(colsum), _stream())
    return out


def gemm_dw(dY, X, dW, accumulate=True, splitk=None):
    """dW[N,K] (+)= dY[M,N]^T X[M,K] (weight gradient), deterministic split-K through a scratch workspace."""
    _chk(dY, bf16, 'dY'), _chk(X, bf16, 'X'), _chk(dW, f32, 'dW')
    M, N = dY.shape
    K = X.shape[1]
    if splitk is None:
        splitk = _lib.load().mmvid_gemm_dw_pick_splitk(M, N, K)
    ws = torch.empty(splitk * N * K, device=dY.device, dtype=f32) if splitk > 1 else None
    call('mmvid_gemm_bf16_dw', M, N, K, _p(dY), N, _p(X), K, splitk, _p(ws), _p(dW), int(accumulate), _stream())
    return dW


def gemm_dw_grouped(dY, X, dWs, accumulate=True):
    """dWs[g][N,K] (+)= dY[g]^T X[g] for all groups in one launch, no split-K (the tower backward's weight gradients of all layers of
    a kind).  dY [G,M,N], X [G,M,K] bf16, contiguous; dWs: list of G fp32 [N,K] tensors or None (skipped)."""
    _chk(dY, bf16, 'dY'), _chk(X, bf16, 'X')
    G, M, N = dY.shape
    K = X.shape[2]
    assert X.shape[:2] == (G, M) and len(dWs) == G
    ptrs = (ctypes.c_void_p * G)(*[(_chk(w, f32, 'dW').data_ptr() if w is not None else None) for w in dWs])
    call('mmvid_gemm_bf16_dw_grouped', M, N, K, _p(dY), dY.stride(1), dY.stride(0), _p(X), X.stride(1), X.stride(0), G, ptrs,
         int(accumulate), _stream())
    return dWs


def _dw_kinds(kinds):
    """[(dY [G,M,N], X [G,M,K], [dW or None] * G)] -> (mmvid_dw_kind_t array, keep-alive list, M, G)."""
    arr = (_lib.DwKind * len(kinds))()
    keep = []
    G, M = kinds[0][0].shape[:2]
    for a, (dY, X, dWs) in zip(arr, kinds):
        _chk(dY, bf16, 'dY'), _chk(X, bf16, 'X')
        assert dY.shape[:2] == (G, M) and X.shape[:2] == (G, M) and len(dWs) == G
        ptrs = (ctypes.c_void_p * G)(*[(_chk(w, f32, 'dW').data_ptr() if w is not None else None) for w in dWs])
        keep.append(ptrs)
        a.N, a.K, a.dY, a.ldy, a.strideY = dY.shape[2], X.shape[2], dY.data_ptr(), dY.stride(1), dY.stride(0)
        a.X, a.ldx, a.strideX, a.dW_list = X.data_ptr(), X.stride(1), X.stride(0), ctypes.cast(ptrs, ctypes.c_void_p)
    return arr, keep, M, G


def gemm_dw_multi(kinds, accumulate=True):
    """Weight gradients of several Linear shapes x several layers in one launch: kinds = [(dY [G,M,N_k], X [G,M,K_k], [dW [N_k,K_k]
    fp32 or None] * G)], at most 4 kinds.  dW (+)= dY[g]^T X[g]."""
    arr, keep, M, G = _dw_kinds(kinds)
    call('mmvid_gemm_bf16_dw_multi', M, len(kinds), arr, G, int(accumulate), _stream())


def gemm_dw_multi_fill(shapes, groups):
    """[(N, K)] -> share of the chip one grouped launch of those weight gradients x `groups` layers keeps busy (host-side policy)."""
    arr = (_lib.DwKind * len(shapes))()
    for a, (N, K) in zip(arr, shapes):
        a.N, a.K, a.dW_list = N, K, None
    return _lib.load().mmvid_gemm_dw_multi_fill(len(shapes), arr, groups)


def gemm_f32(A, B, *, b_kmajor=False, bias=None, residual=None, alpha=1.0):
    """Exact-fp32 GEMM (k-ordered fmaf chains on the f32 matrix pipe): C = alpha * A @ (B if b_kmajor else B^T).
    A [M,K] f32; B [N,K] (row-major) or [K,N] (b_kmajor)."""
    _chk(A, f32, 'A'), _chk(B, f32, 'B')
    M, K = A.shape
    N = B.shape[1] if b_kmajor else B.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=f32)
    call('mmvid_gemm_f32', int(b_kmajor), M, N, K, _p(A), K, _p(B), B.shape[1], 1, 0, 0, 0, float(alpha), _p(bias),
         _p(residual), _p(out), N, _stream())
    return out


# ---------------------------------------------------------------------------------------------- norms
def layernorm_fwd(x, w, b, eps=1e-5, out_dtype=bf16, save_stats=True):
    _chk(x, f32, 'x')
    rows, E = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    mean = torch.empty(rows, device=x.device, dtype=f32) if save_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=f32) if save_stats else None
    call('mmvid_layernorm_fwd', _p(x), E, rows, E, _p(w), _p(b), float(eps), _p(y) if out_dtype == bf16 else None,
         _p(y) if out_dtype == f32 else None, E, _p(mean), _p(rstd), _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, w, dx=None, add=False, dw=None, db=None, dx_colsum=None, workspace=None):
    """workspace (fp32, >= 3*E*64 elements): two-stage deterministic reduction of dw / db / dx_colsum instead of atomics."""
    _chk(dy, f32, 'dy'), _chk(x, f32, 'x')
    rows, E = x.numel() // x.shape[-1], x.shape[-1]
    if dx is None:
        dx = torch.empty_like(x)
        add = False
    call('mmvid_layernorm_bwd_ws', _p(dy), E, _p(x), E, _p(mean), _p(rstd), _p(w), rows, E, _p(dx), E, int(add), None,
         _p(dw), _p(db), _p(dx_colsum), _p(workspace), workspace.numel() if workspace is not None else 0, _stream())
    return dx


def layernorm_bwd_partial(dy, x, mean, rstd, w, workspace, dx=None, add=False, dx_bf16=None, want=(True, True, True)):
    """Stage 1 of the LayerNorm backward: dx (added into `dx` when add) and the partial rows of dw / db / colsum(dx) in `workspace`
    (fp32, this call's own, >= 64 * 3 * E floats).  dy fp32 or bf16.  Returns (dx, blocks)."""
    _chk(x, f32, 'x'), _chk(workspace, f32, 'workspace')
    rows, E = x.numel() // x.shape[-1], x.shape[-1]
    if dx is None:
        dx, add = torch.empty_like(x), False
    nb = ctypes.c_int()
    call('mmvid_layernorm_bwd_partial', _p(dy), int(dy.dtype == bf16), E, _p(x), E, _p(mean), _p(rstd), _p(w), rows, E, _p(dx), E, int(add),
         _p(dx_bf16), int(want[0]), int(want[1]), int(want[2]), _p(workspace), workspace.numel(), ctypes.byref(nb), _stream())
    return dx, nb.value


def layernorm_bwd_reduce_multi(items, blocks, E):
    """items: [(workspace, dw or None, db or None, dx_colsum or None)] -> targets += column sums of each workspace's partial rows."""
    arr = (_lib.LnReduce * len(items))()
    for a, (ws, dw, db, cs) in zip(arr, items):
        a.partial, a.dw, a.db, a.dx_colsum = ws.data_ptr(), _p(dw), _p(db), _p(cs)
    call('mmvid_layernorm_bwd_reduce_multi', len(items), arr, blocks, E, _stream())


def gn_stats_buffer(N, hw, C, device):
    """fp32 scratch of mmvid_groupnorm_swish_nhwc: [N][C][2] affine, then partial sums [N][blocks][32][2] (blocks of 128
    pixels, or of 64 for the strip convolution: sized for the finer one)."""
    return torch.empty(N * (2 * C + 64 * ((hw + 63) // 64)), device=device, dtype=f32)


def conv3x3_strip(x, w, bias, residual=None, out_dtype=bf16, also_bf16=False, gn_stats=None):
    """3x3 / stride 1 / pad 1 on NHWC bf16 in strip form (csrc/conv_strip.hip).  x [N,H,W,Cin], w [Cout,9,Cin] bf16.
    Returns out (and a bf16 copy when out is fp32 and also_bf16).  gn_stats: a gn_stats_buffer (partials per 64 pixels)."""
    _chk(x, bf16, 'x'), _chk(w, bf16, 'w')
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    out = torch.empty(N, H, W, Cout, device=x.device, dtype=out_dtype)
    o16 = torch.empty(N, H, W, Cout, device=x.device, dtype=bf16) if (also_bf16 and out_dtype == f32) else None
    rb = residual if (residual is not None and residual.dtype == bf16) else None
    rf = residual if (residual is not None and residual.dtype == f32) else None
    gp = ctypes.c_void_p(gn_stats.data_ptr() + N * Cout * 2 * 4) if gn_stats is not None else None
    call('mmvid_conv3x3_strip_nhwc', _p(x), N, H, W, Cin, _p(w), _p(bias), Cout, _p(rb), _p(rf),
         _p(out) if out_dtype == bf16 else _p(o16), _p(out) if out_dtype == f32 else None, gp, _stream())
    return (out, o16) if o16 is not None else out


def groupnorm_swish(x, w, b, eps=1e-6, swish=True, out_dtype=bf16, stats=None, stats_block=128):
    """x NHWC [N,H,W,C] bf16 or f32 -> same shape.  `stats`: a gn_stats_buffer whose partial sums were already
    written by the convolution that produced x (conv2d_nhwc(..., gn_stats=...))."""
    N, H, W, C = x.shape
    assert x.is_contiguous() and x.dtype in (bf16, f32)
    blocks = 0
    if stats is None:
        stats = gn_stats_buffer(N, H * W, C, x.device)
    else:
        assert (H * W) % stats_block == 0
        blocks = H * W // stats_block
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    call('mmvid_groupnorm_swish_nhwc', _p(x), int(x.dtype == bf16), N, H * W, C, _p(w), _p(b), float(eps), int(swish),
         _p(stats), blocks, _p(y) if out_dtype == bf16 else None, _p(y) if out_dtype == f32 else None, _stream())
    return y


# ------------------------------------------------------------------------------------------ attention
def _mask_args(mask):
    """mask: None | 'causal' | ('rows', [(row, first_allowed_col), ...])."""
    if mask is None:
        return 0, -1, 0, -1, 0
    if mask == 'causal':
        return 1, -1, 0, -1, 0
    kind, rows = mask
    assert kind == 'rows' and len(rows) <= 2
    rows = list(rows) + [(-1, 0)] * (2 - len(rows))
    return 2, rows[0][0], rows[0][1], rows[1][0], rows[1][1]


def attention_fwd(qkv, B, L, H, mask=None, scale=0.125):
    """qkv [B*L, 3E] bf16 -> (out [B*L, E] bf16, lse2 [B,H,L] f32)."""
    _chk(qkv, bf16, 'qkv')
    E = H * 64
    out = torch.empty(B * L, E, device=qkv.device, dtype=bf16)
    lse2 = torch.empty(B, H, L, device=qkv.device, dtype=f32)
    call('mmvid_attention_fwd', _p(qkv), 3 * E, B, L, H, E, float(scale), *_mask_args(mask), _p(out), E, _p(lse2),
         _stream())
    return out, lse2


def attention_bwd(qkv, out, dout, lse2, B, L, H, mask=None, scale=0.125, dbias=None):
    """-> dqkv [B*L, 3E] bf16; dbias [3E] f32 (optional) += its column sums."""
    E = H * 64
    delta = torch.empty(B, H, L, device=qkv.device, dtype=f32)
    dqkv = torch.empty_like(qkv)
    call('mmvid_attention_bwd_bias', _p(qkv), 3 * E, _p(out), E, _p(dout), E, _p(lse2), _p(delta), B, L, H, E, float(scale),
         *_mask_args(mask), _p(dqkv), 3 * E, _p(dbias) if dbias is not None else None, _stream())
    return dqkv


# --------------------------------------------------------------------------------- sequence / losses
def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else 0 for t in tensors])
    return ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p)), arr


def assemble_sequence(tables, ids, seg, pos):
    """x[b,l,:] = tables[seg[l]][ids[b,l]] + pos[l].  tables: list of [V_i, E] f32; ids [B,L] i64; seg [L] i32."""
    B, L = ids.shape
    E = pos.shape[-1]
    for t in tables:
        _chk(t, f32, 'table')
    _chk(ids, i64, 'ids'), _chk(pos, f32, 'pos')
    assert seg.dtype == torch.int32 and seg.is_cuda
    out = torch.empty(B, L, E, device=ids.device, dtype=f32)
    tp, keep = _ptr_array(tables)
    rows = (ctypes.c_int64 * len(tables))(*[t.shape[0] for t in tables])
    call('mmvid_assemble_sequence', tp, rows, len(tables), _p(ids), _p(seg), _p(pos), B, L, E, _p(out), _stream())
    return out


def assemble_sequence_bwd(grad_tables, table_rows, ids, seg, dx, dpos=None, accumulate_dpos=False):
    B, L = ids.shape
    E = dx.shape[-1]
    _chk(dx, f32, 'dx')
    tp, keep = _ptr_array(grad_tables)
    rows = (ctypes.c_int64 * len(grad_tables))(*table_rows)
    call('mmvid_assemble_sequence_bwd', tp, rows, len(grad_tables), _p(ids), _p(seg), _p(dx), B, L, E, _p(dpos),
         int(accumulate_dpos), _stream())


def cross_entropy_fwd(logits, target, select):
    """Returns (lse [rows], loss_sum [1]) over rows with select != 0 (select uint8 or None)."""
    _chk(logits, f32, 'logits'), _chk(target, i64, 'target')
    rows, V = logits.shape
    lse = torch.empty(rows, device=logits.device, dtype=f32)
    loss = torch.zeros(1, device=logits.device, dtype=f32)
    call('mmvid_cross_entropy_fwd', _p(logits), V, _p(target), _p(select), rows, V, _p(lse), _p(loss), _stream())
    return lse, loss


def cross_entropy_bwd(logits, target, select, lse, gscale):
    rows, V = logits.shape
    d = torch.empty(rows, V, device=logits.device, dtype=bf16)
    call('mmvid_cross_entropy_bwd', _p(logits), V, _p(target), _p(select), _p(lse), _p(gscale), rows, V, _p(d), V,
         _stream())
    return d


def colsum_bf16(dy, db):
    M, N = dy.shape
    call('mmvid_colsum_bf16', _p(dy), N, M, N, _p(db), _stream())


# ------------------------------------------------------------------------------------------ optimiser
def grad_sqnorm(g, out, partials=None, lazy=None):
    """out[0] += sum g^2.  With `partials` (fp32 [2048] scratch) the reduction order is fixed (deterministic).
    lazy = (row_flags uint8 [rows], table_lo, rows, rowlen): skip the rows of that table whose flag is 0 (they are all-zero)."""
    _chk(g, f32, 'g'), _chk(out, f32, 'out')
    if lazy is not None:
        assert partials is not None
        fl, lo, rows, rowlen = lazy
        call('mmvid_grad_sqnorm_rows', _p(g), g.numel(), _p(partials), _p(out), _p(_chk(fl, torch.uint8, 'row_flags')), int(lo), int(rows),
             int(rowlen), _stream())
    elif partials is not None:
        call('mmvid_grad_sqnorm_det', _p(g), g.numel(), _p(partials), _p(out), _stream())
    else:
        call('mmvid_grad_sqnorm', _p(g), g.numel(), _p(out), _stream())


def adam_step(p, g, m, v, shadow, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0,
              sqnorm=None, grad_scale=1.0, step_dev=None, lr_dev=None, lazy=None):
    for t, n in ((p, 'p'), (g, 'g'), (m, 'm'), (v, 'v')):
        _chk(t, f32, n)
    fl, lo, rows, rowlen = lazy if lazy is not None else (None, 0, 0, 0)
    call('mmvid_adam_step_rows', _p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), float(lr), _p(lr_dev), float(betas[0]),
         float(betas[1]), float(eps), float(weight_decay), int(step), _p(step_dev), float(max_norm), _p(sqnorm),
         float(grad_scale), _p(fl), int(lo), int(rows), int(rowlen), _stream())


def lr_schedule(step_dev, kind, lr_min, lr_max, warmup, every, lr_out):
    call('mmvid_lr_schedule', _p(step_dev), int(kind), float(lr_min), float(lr_max), int(warmup), int(every), _p(lr_out),
         _stream())


def counter_add(counter, value=1.0):
    call('mmvid_counter_add', _p(counter), float(value), _stream())


def cast_bf16(x, out=None):
    _chk(x, f32, 'x')
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=bf16)
    call('mmvid_cast_f32_to_bf16', _p(x), _p(out), x.numel(), _stream())
    return out


# ---------------------------------------------------------------------------------------------- VQGAN
def conv2d_nhwc(x, w, bias, mode, residual=None, clamp01=False, out_dtype=bf16, gn_stats=None, splitk=1):
    """x [N,H,W,Cin] bf16; w [Cout,taps,Cin] bf16 (taps 9 or 1); mode 0 3x3 | 1 down | 2 up | 3 1x1.
    gn_stats: a gn_stats_buffer for the OUTPUT shape; its partial-sum area is filled by the epilogue.
    splitk > 1: the reduction is cut into that many ranges, added in a fixed order (deep layers on small maps)."""
    _chk(x, bf16, 'x'), _chk(w, bf16, 'w')
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H // 2, W // 2) if mode == 1 else ((2 * H, 2 * W) if mode == 2 else (H, W))
    out = torch.empty(N, Ho, Wo, Cout, device=x.device, dtype=out_dtype)
    rb = residual if (residual is not None and residual.dtype == bf16) else None
    rf = residual if (residual is not None and residual.dtype == f32) else None
    gp = None
    if gn_stats is not None:
        gp = ctypes.c_void_p(gn_stats.data_ptr() + N * Cout * 2 * 4)
    ws = torch.empty(splitk * N * Ho * Wo * Cout, device=x.device, dtype=f32) if splitk > 1 else None
    call('mmvid_conv2d_nhwc_splitk', mode, _p(x), N, H, W, Cin, _p(w), _p(bias), Cout, _p(rb), _p(rf), int(clamp01),
         _p(out) if out_dtype == bf16 else None, _p(out) if out_dtype == f32 else None, gp, int(splitk), _p(ws), _stream())
    return out


# ---- the split operator (vae.strict = 'split'): bf16 pair planes, three products per convolution -------------------------
def split_planes(x):
    """fp32 [...] -> bf16 pair planes [2, ...]: hi = bf16(x), lo = bf16(x - hi)."""
    _chk(x, f32, 'x')
    out = torch.empty((2, ) + tuple(x.shape), device=x.device, dtype=bf16)
    call('mmvid_split_f32_bf16x2', _p(x.contiguous()), x.numel(), _p(out), _stream())
    return out


def split_weights(w):
    """fp32 [Cout, taps, Cin] -> bf16 [Cout, 3, taps, Cin] = (w_hi | w_hi | w_lo)."""
    hi = w.to(bf16)
    lo = (w - hi.float()).to(bf16)
    return torch.stack([hi, hi, lo], 1).contiguous()


def conv2d_nhwc_split3(x_planes, w3, bias, mode, residual=None, clamp01=False, splitk=1, strip=False, planes_out=None):
    """x_planes [2,N,H,W,Cin] bf16, w3 [Cout,3,taps,Cin] bf16 -> fp32 [N,Ho,Wo,Cout] = conv of (x_hi + x_lo) with (w_hi + w_lo)
    without the lo.lo term, fp32 accumulate (mmvid_conv2d_nhwc_split3 / the strip form for mode 0)."""
    _chk(x_planes, bf16, 'x_planes'), _chk(w3, bf16, 'w3')
    _, N, H, W, Cin = x_planes.shape
    Cout = w3.shape[0]
    Ho, Wo = (H // 2, W // 2) if mode == 1 else ((2 * H, 2 * W) if mode == 2 else (H, W))
    out = torch.empty(N, Ho, Wo, Cout, device=x_planes.device, dtype=f32)
    if strip:
        assert mode == 0 and not clamp01 and splitk == 1
        call('mmvid_conv3x3_strip_nhwc_split3', _p(x_planes), N, H, W, Cin, _p(w3), _p(bias), Cout, _p(residual), _p(out), None, _p(planes_out),
             _stream())
        return out
    ws = torch.empty(splitk * N * Ho * Wo * Cout, device=out.device, dtype=f32) if splitk > 1 else None
    call('mmvid_conv2d_nhwc_split3', mode, _p(x_planes), N, H, W, Cin, _p(w3), _p(bias), Cout, _p(residual), int(clamp01), _p(out),
         None, int(splitk), _p(ws), _stream())
    return out


def groupnorm_swish_split(x, w, b, eps=1e-6, swish=True):
    """x NHWC fp32 -> bf16 pair planes [2,N,H,W,C] of GroupNorm(32)(x) [* sigmoid]."""
    _chk(x, f32, 'x')
    N, H, W, C = x.shape
    out = torch.empty(2, N, H, W, C, device=x.device, dtype=bf16)
    st = torch.empty(N * (2 * C + 64 * ((H * W + 63) // 64)), device=x.device, dtype=f32)
    call('mmvid_groupnorm_swish_nhwc_split', _p(x), N, H * W, C, _p(w), _p(b), float(eps), int(swish), _p(st), 0, _p(out), _stream())
    return out


def groupnorm_swish_f16(x, w, b, eps=1e-6, swish=True):
    """x NHWC fp32 -> fp16 [N,H,W,C] of GroupNorm(32)(x) [* sigmoid] (the input of conv3x3_strip_f16)."""
    _chk(x, f32, 'x')
    N, H, W, C = x.shape
    out = torch.empty(N, H, W, C, device=x.device, dtype=torch.float16)
    st = torch.empty(N * (2 * C + 64 * ((H * W + 63) // 64)), device=x.device, dtype=f32)
    call('mmvid_groupnorm_swish_nhwc_f16out', _p(x), N, H * W, C, _p(w), _p(b), float(eps), int(swish), _p(st), 0, _p(out), _stream())
    return out


def conv3x3_strip_f16(x, w, bias, residual=None, planes_out=None):
    """x fp16 [N,H,W,Cin], w fp16 [Cout,9,Cin] -> fp32 [N,H,W,Cout]: one product of IEEE-half operands, fp32 accumulate (+ fp32 residual)."""
    _chk(x, torch.float16, 'x'), _chk(w, torch.float16, 'w')
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    out = torch.empty(N, H, W, Cout, device=x.device, dtype=f32)
    call('mmvid_conv3x3_strip_nhwc_f16', _p(x), N, H, W, Cin, _p(w), _p(bias), Cout, _p(residual), _p(out), None, _p(planes_out), _stream())
    return out


def image_to_nhwc8(img):
    _chk(img, f32, 'img')
    N, C, H, W = img.shape
    assert C == 3
    out = torch.empty(N, H, W, 8, device=img.device, dtype=bf16)
    call('mmvid_image_to_nhwc8', _p(img), N, H, W, _p(out), _stream())
    return out


def nhwc_to_nchw(x, c_use=None):
    _chk(x, f32, 'x')
    N, H, W, C = x.shape
    c_use = c_use or C
    out = torch.empty(N, c_use, H, W, device=x.device, dtype=f32)
    call('mmvid_nhwc_to_nchw_f32', _p(x), N, H, W, C, c_use, _p(out), _stream())
    return out


def spatial_attention(q, k, v):
    """q,k,v [N,HW,C] bf16 -> softmax(q k^T C^-0.5) v, bf16."""
    N, HW, C = q.shape
    scratch = torch.empty(N * HW * HW * 3 // 2 + 16, device=q.device, dtype=f32)
    out = torch.empty_like(q)
    call('mmvid_spatial_attention', _p(q), _p(k), _p(v), N, HW, C, float(C)**-0.5, _p(scratch), _p(out), _stream())
    return out


# ------------------------------------------------------------------------------------------- samplers
u8 = torch.uint8


def exponential_like(shape, device):
    """Exp(1) race variates for the samplers (what torch.multinomial draws internally), from torch's device generator."""
    return torch.empty(shape, device=device, dtype=f32).exponential_()


def sample_race(logits, E, noise_u=None, temperature=0.0, logit_div=1.0, tok_offset=0, want_y=True, tok_out=None, step_dev=None, step0=0):
    """logits [R, V] f32 (row stride = logits.stride(0)), E [R, V] f32 Exp(1) variates -> (tok int64 [R] (tok_out if given), y f32 [R]).
    step_dev (int32 device scalar): E is [draws, R, V] and draw number step_dev - step0 is used (token-only form: want_y False, no noise)."""
    _chk(E, f32, 'E')
    assert logits.dtype == f32 and logits.is_cuda and logits.stride(1) == 1
    R, V = logits.shape
    assert E.shape[-2:] == (R, V) and E.is_contiguous() and (E.dim() == 2 or (E.dim() == 3 and step_dev is not None))
    if tok_out is not None:
        _chk(tok_out, i64, 'tok_out')
        assert tok_out.shape == (R, )
    tok = tok_out if tok_out is not None else torch.empty(R, device=logits.device, dtype=i64)
    y = torch.empty(R, device=logits.device, dtype=f32) if want_y else None
    if noise_u is not None:
        _chk(noise_u, f32, 'noise_u')
    call('mmvid_sample_race_at', _p(logits), logits.stride(0), _p(E), _p(step_dev) if step_dev is not None else None, int(step0), R * V,
         _p(noise_u), float(temperature), float(logit_div), R, V, int(tok_offset), _p(tok), _p(y), _stream())
    return tok, y


def mp_select_keep(Y, E, preserve, k):
    """Y [b, TS], E [b, Bm, TS], preserve [TS] uint8 or None -> mask1 [b, Bm, TS] uint8 (1 = position stays visible)."""
    _chk(Y, f32, 'Y'), _chk(E, f32, 'E')
    b, Bm, TS = E.shape
    mask1 = torch.empty(b, Bm, TS, device=Y.device, dtype=u8)
    call('mmvid_mp_select_keep', _p(Y), _p(E), _p(preserve), b, Bm, TS, int(k), _p(mask1), _stream())
    return mask1


def mp_build_input(control_emb, image_emb, tpos, I_tok, mask1, Bm, mask_id):
    _chk(control_emb, f32, 'control_emb'), _chk(image_emb, f32, 'image_emb'), _chk(tpos, f32, 'tpos'), _chk(I_tok, i64, 'I_tok')
    b, csl, E = control_emb.shape
    TS = I_tok.shape[1]
    out = torch.empty(b * Bm, csl + TS, E, device=control_emb.device, dtype=f32)
    call('mmvid_mp_build_input', _p(control_emb), _p(image_emb), image_emb.shape[0], _p(tpos), _p(I_tok), _p(mask1), b, Bm, csl,
         TS, E, int(mask_id), _p(out), _stream())
    return out


def mp_update(mask1, Ynew, Inew, rel_logit, vid_logit, t, dynamic, Y, I_tok, Imax, Smax, tmax, active, S_out=None,
              jmax_out=None):
    b, Bm, TS = mask1.shape
    call('mmvid_mp_update', _p(mask1), _p(Ynew), _p(Inew), _p(rel_logit), _p(vid_logit), b, Bm, TS, int(t), int(dynamic),
         _p(Y), _p(I_tok), _p(Imax), _p(Smax), _p(tmax), _p(active), _p(S_out), _p(jmax_out), _stream())


def head_rows_fwd(x2d, rows, ln_w, ln_b, w, b, eps=1e-5, label=None, row_weight=None, den_from=None, den_const=1.0):
    """z[r] = LN(x2d[rows[r]]) . w + b (the 768 -> 1 heads); with `label` also the BCE loss [1].  Returns
    (z, mean, rstd, loss|None)."""
    _chk(x2d, f32, 'x'), _chk(rows, i64, 'rows')
    R, E = rows.numel(), x2d.shape[1]
    dev = x2d.device
    z, mean, rstd = (torch.empty(R, device=dev, dtype=f32) for _ in range(3))
    loss = torch.empty(1, device=dev, dtype=f32) if label is not None else None
    call('mmvid_head_bce_fwd', _p(x2d), x2d.stride(0), _p(rows), R, E, _p(ln_w), _p(ln_b), float(eps), _p(w), _p(b),
         _p(label), _p(row_weight), _p(den_from), den_from.numel() if den_from is not None else 0, float(den_const), _p(z),
         _p(mean), _p(rstd), _p(loss), _stream())
    return z, mean, rstd, loss


def head_rows_bwd(x2d, rows, ln_w, ln_b, w, z, mean, rstd, label, row_weight, den_from, den_const, gloss, dx2d, dw, db,
                  dln_w, dln_b):
    R, E = rows.numel(), x2d.shape[1]
    call('mmvid_head_bce_bwd', _p(x2d), x2d.stride(0), _p(rows), R, E, _p(ln_w), _p(ln_b), _p(w), _p(z), _p(mean), _p(rstd),
         _p(label), _p(row_weight), _p(den_from), den_from.numel() if den_from is not None else 0, float(den_const),
         _p(gloss), _p(dx2d), dx2d.stride(0), _p(dw), _p(db), _p(dln_w), _p(dln_b), _stream())


def bert_build_ids(text, visual_tok, nvis, target, target_warp, mask1, pad_base, mask_id, has_rel, has_vid, text_neg=None):
    """-> (ids [(1+rel+vid)*B, L] int64, select_full [B*L] uint8, target_full [B*L] int64, select_count [1] f32)."""
    _chk(text, i64, 'text'), _chk(target, i64, 'target')
    B, Ttxt = text.shape
    TS = target.shape[1]
    Nvis = int(nvis)  # visual_tok None with nvis > 0: the visual segment is all [MASK] (dalle_bert.py:954-957)
    assert visual_tok is None or visual_tok.shape == (B, Nvis)
    L = 1 + Ttxt + Nvis + 2 + TS
    nseq = 1 + int(has_rel) + int(has_vid)
    dev = text.device
    ids = torch.empty(nseq * B, L, device=dev, dtype=i64)
    sel = torch.empty(B * L, device=dev, dtype=u8)
    tfull = torch.empty(B * L, device=dev, dtype=i64)
    cnt = torch.empty(1, device=dev, dtype=f32)
    m1 = mask1 if mask1.dtype == u8 else mask1.to(u8)
    call('mmvid_bert_build_ids', _p(text), _p(text_neg), _p(visual_tok), _p(target), _p(target_warp), _p(m1.contiguous()), B,
         Ttxt, Nvis, TS, int(pad_base), int(mask_id), int(has_rel), int(has_vid), _p(ids), _p(sel), _p(tfull), _p(cnt),
         _stream())
    return ids, sel, tfull, cnt


def gemv_rows(x, W, bias=None, ln=None, act=0, residual=None, round_in=False, round_out=False, out=None):
    """Decode-time linear layer on a few rows: y = act(LN?(x) @ W^T + bias) (+ residual).  x [NB, K] f32 (contiguous), W [N, K] bf16."""
    _chk(x, f32, 'x'), _chk(W, bf16, 'W')
    NB, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(NB, N, device=x.device, dtype=f32)
    lw, lb, eps = (ln[0], ln[1], ln[2]) if ln is not None else (None, None, 0.0)
    step = 64 if (round_in and K <= 1024 and K % 256 == 0 and N % 16 == 0) else (16 if round_in else 8)  # rows per launch: 64 bf16-exact rows on the MFMA form, 16 / 8 on the vector-ALU kernel; larger batches in slices
    for r0 in range(0, NB, step):
        nb = min(step, NB - r0)
        res = residual[r0:r0 + nb] if residual is not None else None
        call('mmvid_gemv_rows', _p(x[r0:r0 + nb]), K, nb, K, _p(lw), _p(lb), float(eps), _p(W), _p(bias), N, int(act), _p(res), N,
             int(round_in), int(round_out), _p(out[r0:r0 + nb]), N, _stream())
    return out


def decode_embed(tok, table, pos_rows, pos_dev, out, pos_off=0, record=None, record_pos0=0):
    """out[b] = table[tok[b]] + pos_rows[pos_dev + pos_off] (the embedding row of a freshly sampled token); record (int64 [B, n],
    optional): record[b, pos_dev - record_pos0] = tok[b]."""
    _chk(tok, i64, 'tok'), _chk(table, f32, 'table'), _chk(pos_rows, f32, 'pos_rows')
    B, E = out.shape
    if record is not None:
        _chk(record, i64, 'record')
        assert record.shape[0] == B and record.stride(1) == 1
    call('mmvid_decode_embed_record', _p(tok), _p(table), table.shape[0], _p(pos_rows), _p(pos_dev), int(pos_off), B, E, _p(out),
         _p(record) if record is not None else None, record.stride(0) if record is not None else 0, int(record_pos0), _stream())
    return out
