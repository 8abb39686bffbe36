"""autograd glue (plumbing) between the boundary modules and the HIP kernels: each Function's forward and
backward are C-ABI calls; parameter gradients are accumulated by the kernels straight into `p.grad`."""
import torch

from . import ops

bf16, f32 = torch.bfloat16, torch.float32


def _grad_buf(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class AssembleSequence(torch.autograd.Function):
    """x[b,l,:] = tables[seg[l]][ids[b,l]] + pos[l]   (dalle_bert.py:899-973, 1030-1035; dalle_artv.py:441-491).
    `pos` is a dense [L, E] tensor built from the (tiny) positional parameters with ordinary torch ops, so its
    gradient flows back through autograd; table gradients are scatter-added by the kernel."""

    @staticmethod
    def forward(ctx, pos, ids, seg, *tables):
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
        return (dpos, None, None) + (None, ) * len(ctx.tables)


class LNLinear(torch.autograd.Function):
    """nn.Sequential(nn.LayerNorm(E), nn.Linear(E, N)) of dalle_bert.py:414-417 / dalle_artv.py:210-213 on rows
    x [R, E] fp32 -> logits [R, N] fp32 (bf16 MFMA GEMM, fp32 accumulate)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, b, w_bf16):
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
        return LNLinear._backward_bf16(d16, x, mean, rstd, h, ln_w, ln_b, w, b, w_bf16)

    @staticmethod
    def _backward_bf16(d16, x, mean, rstd, h, ln_w, ln_b, w, b, w_bf16):
        R, N = d16.shape
        E = x.shape[1]
        if w.requires_grad:
            ops.gemm_dw(d16, h, _grad_buf(w), accumulate=True)
        if b.requires_grad:
            ops.colsum_bf16(d16, _grad_buf(b))
        dh = ops.gemm(d16, w_bf16, b_kmajor=True, out_dtype=f32)  # [R, E]
        dx = ops.layernorm_bwd(dh, x, mean, rstd, ln_w.detach(), dw=_grad_buf(ln_w) if ln_w.requires_grad else None,
                               db=_grad_buf(ln_b) if ln_b.requires_grad else None)
        return dx, None, None, None, None, None


class LNLinearCrossEntropy(torch.autograd.Function):
    """to_logits + F.cross_entropy(logits[select], target[select]) (dalle_bert.py:1038-1040) fused at the autograd
    level: the bf16 dlogits go straight into the backward GEMMs.  Returns (loss, logits)."""

    @staticmethod
    def forward(ctx, x, target, select, ln_w, ln_b, w, b, w_bf16):
        x = x.contiguous()
        h, mean, rstd = ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), 1e-5)
        logits = ops.gemm(h, w_bf16, bias=b.detach(), out_dtype=f32)
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
        dx = LNLinear._backward_bf16(d16, x, mean, rstd, h, *ctx.params)[0]
        return (dx, ) + (None, ) * 7
