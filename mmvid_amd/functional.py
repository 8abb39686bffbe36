"""autograd glue (plumbing) between the boundary modules and the HIP kernels: each Function's forward and
backward are C-ABI calls; parameter gradients are accumulated by the kernels straight into `p.grad`."""
import torch

from . import ops

bf16, f32 = torch.bfloat16, torch.float32


def _grad_buf(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


# ---- torch.nn.parallel.DistributedDataParallel compatibility (train.py:28-35 wraps the model in DDP) -----------------------
# The kernels add parameter gradients straight into `p.grad`, so autograd sees no gradient for a parameter and DDP's reducer --
# which hooks every parameter's gradient accumulation -- would wait for hooks that never fire.  When a Function's forward runs
# inside a DDP-wrapped module's forward, its backward therefore returns a (stride-0, scalar-backed) ZERO gradient for every
# trainable parameter input: autograd's accumulation `p.grad += 0` then runs after the kernels have written the real gradient,
# the hook fires, and DDP all-reduces `p.grad` as usual.  Cost: one extra elementwise pass per parameter per step; the flat
# engine (engine.FlatTrainer) avoids both DDP and this.
_ZERO = {}


def _inside_ddp():
    ddp = getattr(torch.nn.parallel, 'DistributedDataParallel', None)
    return ddp is not None and getattr(ddp, '_active_ddp_module', None) is not None


def _note_params(ctx, args):
    """Call in forward: remembers which inputs are trainable parameters when the model runs under DDP."""
    ctx._ddp_params = [a if (isinstance(a, torch.nn.Parameter) and a.requires_grad) else None for a in args] if _inside_ddp() else None


def _with_param_zeros(ctx, grads):
    """Call on backward's return tuple: under DDP, None -> zeros for the parameter inputs noted in forward."""
    ps = getattr(ctx, '_ddp_params', None)
    if ps is None:
        return grads
    out = list(grads)
    for i, p in enumerate(ps):
        if p is not None and i < len(out) and out[i] is None:
            _grad_buf(p)  # exists (zeros) even if no kernel of this backward wrote it: the accumulation below stays in place
            key = (p.device, p.dtype)
            if key not in _ZERO:
                _ZERO[key] = torch.zeros((), device=p.device, dtype=p.dtype)
            out[i] = _ZERO[key].expand(p.shape)
    return tuple(out)


class AssembleSequence(torch.autograd.Function):
    """x[b,l,:] = tables[seg[l]][ids[b,l]] + pos[l]   (dalle_bert.py:899-973, 1030-1035; dalle_artv.py:441-491).
    `pos` is a dense [L, E] tensor built from the (tiny) positional parameters with ordinary torch ops, so its
    gradient flows back through autograd; table gradients are scatter-added by the kernel."""

    @staticmethod
    def forward(ctx, pos, ids, seg, *tables):
        _note_params(ctx, (pos, ids, seg) + tables)
        ctx.tables = tables
        ctx.save_for_backward(ids, seg)
        return ops.assemble_sequence([t.detach() for t in tables], ids, seg, pos.detach().contiguous())

    @staticmethod
    def backward(ctx, dx):
        ids, seg = ctx.saved_tensors
        dx = dx.contiguous()
        gts = [_grad_buf(t) if t.requires_grad else None for t in ctx.tables]
        dpos = torch.empty(dx.shape[1:], device=dx.device, dtype=f32) if ctx.needs_input_grad[0] else None
        ops.assemble_sequence_bwd(gts, [t.shape[0] for t in ctx.tables], ids, seg, dx, dpos)
        return _with_param_zeros(ctx, (dpos, None, None) + (None, ) * len(ctx.tables))


class PosTable(torch.autograd.Function):
    """The dense positional table [L, E] of a sequence from its (tiny) parameters in ONE launch, gradients accumulated into
    `p.grad` in ONE launch (the torch construction -- slices, expands, sums, cat -- and its backward were 22 of the step's
    41 framework launches).  `layout`: list of (dst0, rows, src0, (params...), dims) -- see mmvid_pos_table_fwd."""

    @staticmethod
    def forward(ctx, layout, L, *params):
        from . import _lib
        _note_params(ctx, (layout, L) + params)
        E = params[0].shape[-1]
        ctx.layout, ctx.L, ctx.E = layout, L, E
        out = torch.empty(L, E, device=params[0].device, dtype=f32)
        segs = PosTable._segments(layout, grads=False)
        _lib.call('mmvid_pos_table_fwd', segs, len(layout), L, E, ops._p(out), ops._stream())
        return out

    @staticmethod
    def _segments(layout, grads):
        from . import _lib
        arr = (_lib.PosSegment * len(layout))()
        for k, (dst0, rows, src0, ws, dims) in enumerate(layout):
            sg = arr[k]
            sg.dst0, sg.rows, sg.src0, sg.naxes = dst0, rows, src0, len(dims)
            for a in range(3):
                w = ws[a] if a < len(ws) else None
                sg.w[a] = w.data_ptr() if w is not None else None
                sg.gw[a] = _grad_buf(w).data_ptr() if (grads and w is not None and w.requires_grad) else None
                sg.d[a] = dims[a] if a < len(dims) else 1
        return arr

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        g = ops._chk(g.contiguous(), f32, 'dpos')
        segs = PosTable._segments(ctx.layout, grads=True)
        _lib.call('mmvid_pos_table_bwd', segs, len(ctx.layout), ctx.E, ops._p(g), ops._stream())
        return _with_param_zeros(ctx, (None, None) + (None, ) * (len(ctx.needs_input_grad) - 2))


class WeightedLoss(torch.autograd.Function):
    """wa * a + wb * b + wc * c on device scalars (train.py:320) as one launch forward, one backward."""

    @staticmethod
    def forward(ctx, wa, wb, wc, a, b, c):
        from . import _lib
        ctx.w = (float(wa), float(wb), float(wc))
        out = torch.empty((), device=a.device, dtype=f32)
        sc = [ops._chk(t.detach().reshape(1).contiguous(), f32, 'loss') for t in (a, b, c)]
        _lib.call('mmvid_lincomb3', ops._p(sc[0]), ops._p(sc[1]), ops._p(sc[2]), *ctx.w, ops._p(out), ops._stream())
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        gs = torch.empty(3, device=g.device, dtype=f32)
        g = g.to(f32).reshape(1).contiguous()
        _lib.call('mmvid_scale3', ops._p(g), *ctx.w, ops._p(gs[0:1]), ops._p(gs[1:2]), ops._p(gs[2:3]), ops._stream())
        return None, None, None, gs[0].reshape(()), gs[1].reshape(()), gs[2].reshape(())


def weighted_loss(losses, weights):
    """sum_i weights[i] * losses[i] for the three BERT losses (train.py:320), fused."""
    lm, lr, lv = losses
    return WeightedLoss.apply(weights[0], weights[1], weights[2], lm, lr, lv)


class LNLinear(torch.autograd.Function):
    """nn.Sequential(nn.LayerNorm(E), nn.Linear(E, N)) of dalle_bert.py:414-417 / dalle_artv.py:210-213 on rows
    x [R, E] fp32 -> logits [R, N] fp32 (bf16 MFMA GEMM, fp32 accumulate)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, b, w_bf16):
        _note_params(ctx, (x, ln_w, ln_b, w, b, w_bf16))
        x = x.contiguous()
        h, mean, rstd = ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), 1e-5)
        y = ops.gemm(h, w_bf16, bias=b.detach(), out_dtype=f32)
        ctx.save_for_backward(x, mean, rstd, h)
        ctx.params = (ln_w, ln_b, w, b, w_bf16)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, h = ctx.saved_tensors
        ln_w, ln_b, w, b, w_bf16 = ctx.params
        d16 = dy if dy.dtype == bf16 else ops.cast_bf16(dy.contiguous())
        return _with_param_zeros(ctx, LNLinear._backward_bf16(d16, x, mean, rstd, h, ln_w, ln_b, w, b, w_bf16))

    @staticmethod
    def _backward_bf16(d16, x, mean, rstd, h, ln_w, ln_b, w, b, w_bf16, cols=None):
        R, N = d16.shape
        E = x.shape[1]
        c0, c1 = cols if cols is not None else (0, w.shape[0])
        if w.requires_grad:
            ops.gemm_dw(d16, h, _grad_buf(w)[c0:c1], accumulate=True)
        if b.requires_grad:
            ops.colsum_bf16(d16, _grad_buf(b)[c0:c1])
        dh = ops.gemm(d16, w_bf16[c0:c1], b_kmajor=True, out_dtype=f32)  # [R, E]
        dx = ops.layernorm_bwd(dh, x, mean, rstd, ln_w.detach(), dw=_grad_buf(ln_w) if ln_w.requires_grad else None,
                               db=_grad_buf(ln_b) if ln_b.requires_grad else None)
        return dx, None, None, None, None, None


class Linear(torch.autograd.Function):
    """nn.Linear(K, N) on rows x [R, K] fp32 -> [R, N] fp32 (bf16 MFMA GEMM, fp32 accumulate): the `text_feature_mapping` of the
    fixed-language-model branch (dalle_bert.py:322)."""

    @staticmethod
    def forward(ctx, x, w, b, w_bf16):
        _note_params(ctx, (x, w, b, w_bf16))
        x16 = ops.cast_bf16(x.contiguous())
        ctx.save_for_backward(x16)
        ctx.params = (w, b, w_bf16)
        return ops.gemm(x16, w_bf16, bias=b.detach(), out_dtype=f32)

    @staticmethod
    def backward(ctx, dy):
        x16, = ctx.saved_tensors
        w, b, w_bf16 = ctx.params
        d16 = ops.cast_bf16(dy.contiguous())
        if w.requires_grad:
            ops.gemm_dw(d16, x16, _grad_buf(w), accumulate=True)
        if b.requires_grad:
            ops.colsum_bf16(d16, _grad_buf(b))
        dx = ops.gemm(d16, w_bf16, b_kmajor=True, out_dtype=f32) if ctx.needs_input_grad[0] else None
        return _with_param_zeros(ctx, (dx, None, None, None))


class LayerNormRows(torch.autograd.Function):
    """nn.LayerNorm(E) on rows [R, E] fp32 -> fp32 (the closing norm of the bottleneck mapping, dalle_bert.py:319)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _note_params(ctx, (x, w, b))
        x = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x, w.detach(), b.detach(), 1e-5, out_dtype=f32)
        ctx.save_for_backward(x, mean, rstd)
        ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        w, b = ctx.params
        dx = ops.layernorm_bwd(dy.contiguous(), x, mean, rstd, w.detach(), dw=_grad_buf(w) if w.requires_grad else None,
                               db=_grad_buf(b) if b.requires_grad else None)
        return _with_param_zeros(ctx, (dx, None, None))


class LNLinearCrossEntropy(torch.autograd.Function):
    """to_logits + F.cross_entropy(logits[select], target[select]) (dalle_bert.py:1038-1040) fused at the autograd
    level: the bf16 dlogits go straight into the backward GEMMs.  Returns (loss, logits).
    cols = (c0, c1): only that block of output classes exists for these rows (ART-V's block-diagonal vocabulary mask,
    dalle_artv.py:215-227, 509-512: every other class has logit -max, i.e. probability exactly 0); `target` is then
    relative to c0 and the weight / bias gradients go to rows c0:c1 of the full parameters."""

    @staticmethod
    def forward(ctx, x, target, select, ln_w, ln_b, w, b, w_bf16, cols=None):
        _note_params(ctx, (x, target, select, ln_w, ln_b, w, b, w_bf16, cols))
        x = x.contiguous()
        h, mean, rstd = ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), 1e-5)
        ctx.cols = cols
        w_blk = w_bf16 if cols is None else w_bf16[cols[0]:cols[1]]
        bias = b.detach() if cols is None else b.detach()[cols[0]:cols[1]].contiguous()
        logits = ops.gemm(h, w_blk, bias=bias, out_dtype=f32)
        sel8 = select.to(torch.uint8).contiguous() if select is not None else None
        lse, loss_sum = ops.cross_entropy_fwd(logits, target, sel8)
        cnt = (select.sum() if select is not None else torch.tensor(logits.shape[0], device=x.device)).to(f32)
        ctx.save_for_backward(x, mean, rstd, h, logits, target, sel8, lse, cnt)
        ctx.params = (ln_w, ln_b, w, b, w_bf16)
        ctx.mark_non_differentiable(logits)
        return (loss_sum / cnt).squeeze(0), logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        x, mean, rstd, h, logits, target, sel8, lse, cnt = ctx.saved_tensors
        gs = (gloss.to(f32) / cnt).reshape(1).contiguous()
        d16 = ops.cross_entropy_bwd(logits, target, sel8, lse, gs)
        dx = LNLinear._backward_bf16(d16, x, mean, rstd, h, *ctx.params, cols=ctx.cols)[0]
        return _with_param_zeros(ctx, (dx, ) + (None, ) * 8)


class BertHeads(torch.autograd.Function):
    """All three heads and losses of a BERT training step (dalle_bert.py:1038-1040, 1062-1084, 1103-1123) on the tower
    output y [nseq*B, L, E], forward and backward as HIP launches:
      MSM   to_logits (LayerNorm + MFMA GEMM) on the B*L rows of the first pass and cross entropy over the rows
            `select` marks (the masked target positions; control rows are never selected, so their logits cost ~13 %
            of a small GEMM and buy a copy-free, uniformly strided row set);
      REL / VID   LayerNorm + 768->1 dot + BCE-with-logits on 2*B gathered rows each (csrc/sample.hip).
    The backward writes ONE gradient tensor for y: the MSM LayerNorm backward fills the first pass's rows, the other
    passes are zero-filled, the small heads add their rows.  Returns (loss_msm, loss_rel, loss_vid, logits [B*L, V])."""

    @staticmethod
    def forward(ctx, y, target_full, select_full, count, nfm, labels, rel_rows, vid_rows, weight_by_nfm, B, w_bf16, ln_w,
                ln_b, w, b, rel_ln_w, rel_ln_b, rel_w, rel_b, vid_ln_w, vid_ln_b, vid_w, vid_b):
        _note_params(ctx, (y, target_full, select_full, count, nfm, labels, rel_rows, vid_rows, weight_by_nfm, B, w_bf16, ln_w,
                           ln_b, w, b, rel_ln_w, rel_ln_b, rel_w, rel_b, vid_ln_w, vid_ln_b, vid_w, vid_b))
        nB, L, E = y.shape
        y2d = y.contiguous().view(nB * L, E)
        rows = B * L
        h, mean, rstd = ops.layernorm_fwd(y2d[:rows], ln_w.detach(), ln_b.detach(), 1e-5)
        logits = ops.gemm(h, w_bf16, bias=b.detach(), out_dtype=f32)
        lse, loss_sum = ops.cross_entropy_fwd(logits, target_full, select_full)
        loss_msm = (loss_sum / count).squeeze(0)
        zero = torch.zeros((), device=y.device, dtype=f32)
        den_from = nfm if weight_by_nfm else None
        saved_small = []
        losses = []
        for rws, lw, lb, hw, hb, per_row in ((rel_rows, rel_ln_w, rel_ln_b, rel_w, rel_b, True),
                                             (vid_rows, vid_ln_w, vid_ln_b, vid_w, vid_b, False)):
            if rws is None:
                losses.append(zero)
                saved_small.append(None)
                continue
            # REL weights each sample's two terms by not_fully_masked (1067-1079); VID only divides by its sum (1107-1116)
            rw = nfm.repeat(2) if (weight_by_nfm and per_row) else None
            z, mu, rs, loss = ops.head_rows_fwd(y2d, rws, lw.detach(), lb.detach(), hw.detach().view(-1), hb.detach(), 1e-5,
                                                label=labels, row_weight=rw, den_from=den_from, den_const=float(B))
            losses.append(loss.squeeze(0))
            saved_small.append((rws, z, mu, rs, rw))
        ctx.save_for_backward(y2d, mean, rstd, h, logits, target_full, select_full, lse, count, labels)
        ctx.small = saved_small
        ctx.den_from, ctx.B = den_from, B
        ctx.params = (w_bf16, ln_w, ln_b, w, b, rel_ln_w, rel_ln_b, rel_w, rel_b, vid_ln_w, vid_ln_b, vid_w, vid_b)
        ctx.shape = (nB, L, E)
        ctx.mark_non_differentiable(logits)
        return loss_msm, losses[0], losses[1], logits

    @staticmethod
    def backward(ctx, g_msm, g_rel, g_vid, _g_logits):
        y2d, mean, rstd, h, logits, target_full, select_full, lse, count, labels = ctx.saved_tensors
        w_bf16, ln_w, ln_b, w, b, rel_ln_w, rel_ln_b, rel_w, rel_b, vid_ln_w, vid_ln_b, vid_w, vid_b = ctx.params
        nB, L, E = ctx.shape
        rows = ctx.B * L
        gy = torch.empty_like(y2d)
        # ---- MSM: dlogits (bf16) -> dW, db, dh -> LayerNorm backward straight into gy's first rows
        gs = (g_msm.to(f32) / count).reshape(1).contiguous()
        d16 = ops.cross_entropy_bwd(logits, target_full, select_full, lse, gs)
        if w.requires_grad:
            ops.gemm_dw(d16, h, _grad_buf(w), accumulate=True)
        if b.requires_grad:
            ops.colsum_bf16(d16, _grad_buf(b))
        dh = ops.gemm(d16, w_bf16, b_kmajor=True, out_dtype=f32)
        ops.layernorm_bwd(dh, y2d[:rows], mean, rstd, ln_w.detach(), dx=gy[:rows],
                          dw=_grad_buf(ln_w) if ln_w.requires_grad else None,
                          db=_grad_buf(ln_b) if ln_b.requires_grad else None)
        if nB * L > rows:
            gy[rows:].zero_()
        # ---- REL / VID rows (added on top)
        for saved, g, (lw, lb, hw, hb), per_row in ((ctx.small[0], g_rel, (rel_ln_w, rel_ln_b, rel_w, rel_b), True),
                                                    (ctx.small[1], g_vid, (vid_ln_w, vid_ln_b, vid_w, vid_b), False)):
            if saved is None or g is None:
                continue
            rws, z, mu, rs, rw = saved
            ops.head_rows_bwd(y2d, rws, lw.detach(), lb.detach(), hw.detach().view(-1), z, mu, rs, labels, rw, ctx.den_from,
                              float(ctx.B), g.to(f32).reshape(1).contiguous(), gy,
                              _grad_buf(hw).view(-1) if hw.requires_grad else None, _grad_buf(hb) if hb.requires_grad else None,
                              _grad_buf(lw) if lw.requires_grad else None, _grad_buf(lb) if lb.requires_grad else None)
        return _with_param_zeros(ctx, (gy.view(nB, L, E), ) + (None, ) * 22)
