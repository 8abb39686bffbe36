"""CPU restatement of the samplers with the randomness injected as Exp(1) race variates (TEST INFRASTRUCTURE).

Reference: BERT mask-predict, mmvid_pytorch/dalle_bert.py:514-714, and the ART-V token draw, dalle_artv.py:61-67,
274-281.  The reference samples with torch.multinomial, whose algorithm is an exponential race: q ~ Exp(1) per
category, take the categories with the largest p / q (equivalently the smallest q / p) -- one of them with
replacement, k of them without.  torch's generator cannot be replayed on another device, so parity is defined on the
race variates instead: oracle and product receive the SAME tensors E and must take identical decisions
(SURVEY section 8c vi, "trajectory with RNG-injected uniforms").  `oracle/bert.py::mask_predict` keeps the
torch-generator form that is pinned against the reference's golden trajectory; tests/test_oracle_golden.py checks that
the two forms sample the same distributions.

Plain numpy, fp32 arithmetic written out (one IEEE operation per step) so the product kernels can match it bit for bit
wherever no transcendental is involved.
"""
import numpy as np

f32 = np.float32


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def gumbel_from_u(u):
    """dalle_bert.py:536-538: -log(-log(U + eps) + eps), eps = 1e-20."""
    u = _np(u).astype(f32)
    return -np.log(-np.log(u + f32(1e-20)) + f32(1e-20))


def token_race(logits, E, temperature=0.0, noise_u=None, logit_div=1.0):
    """logits [R, V], E [R, V] -> (tok int64 [R], Y f32 [R], P f32 [R, V]).  P = exp(x - max x) with
    x = logits * (1 / logit_div) (+ temperature * gumbel(noise_u)); tok = first argmin E / P; Y = P[tok] / sum P."""
    x = _np(logits).astype(f32) * f32(1.0 / logit_div)
    if noise_u is not None:
        x = x + f32(temperature) * gumbel_from_u(noise_u)
    x = x.astype(f32)
    P = np.exp(x - x.max(axis=1, keepdims=True)).astype(f32)
    E = _np(E).astype(f32)
    with np.errstate(divide='ignore'):
        key = np.where(P > 0, E / np.where(P > 0, P, 1), np.inf).astype(f32)
    tok = key.argmin(axis=1)  # first minimum
    Y = (P[np.arange(len(tok)), tok] / P.sum(axis=1, dtype=f32)).astype(f32)
    return tok.astype(np.int64), Y, P


def keep_race(Y, E, preserve, k):
    """Y [TS], E [TS], preserve bool [TS] or None -> keep bool [TS] (True = stays visible): the k valid positions with the
    smallest E / Y (ties: lower index) plus every preserved one.  k outside [1, #valid] or more than the non-zero
    weights: 1 (torch.multinomial raises there and the reference falls back to one sample, dalle_bert.py:653-661)."""
    Y, E = _np(Y).astype(f32), _np(E).astype(f32)
    TS = Y.shape[0]
    valid = np.ones(TS, bool) if preserve is None else ~_np(preserve).astype(bool)
    nz = valid & (Y > 0)
    key = np.full(TS, np.inf, f32)
    key[nz] = E[nz] / Y[nz]
    k_eff = k if (1 <= k <= valid.sum() and k <= nz.sum()) else 1
    order = np.argsort(key, kind='stable')
    keep = ~valid
    chosen = [i for i in order[:k_eff] if np.isfinite(key[i])]
    keep[chosen] = True
    return keep


def update(Y, I_tok, masks, Ynew, Inew, z_rel, z_vid):
    """dalle_bert.py:675-692 for one video: masks / Ynew / Inew [Bm, TS] (masks already include the preserved positions),
    z_* [Bm] head logits -> (Y, I_tok of the best candidate, S [Bm], jmax).  The where-chain is sequential: candidate
    j's update starts from candidate j-1's result."""
    Y, I_tok = _np(Y).astype(f32).copy(), _np(I_tok).copy()
    masks, Ynew, Inew = _np(masks).astype(bool), _np(Ynew).astype(f32), _np(Inew)
    zr, zv = _np(z_rel).astype(np.float64), _np(z_vid).astype(np.float64)
    S = (0.5 / (1 + np.exp(-zr)) + 0.5 / (1 + np.exp(-zv)))
    YB, IB = [], []
    for j in range(masks.shape[0]):
        Y = np.where(masks[j], Y, Ynew[j])
        I_tok = np.where(masks[j], I_tok, Inew[j])
        YB.append(Y), IB.append(I_tok)
    jmax = int(S.argmax())
    return YB[jmax], IB[jmax], S, jmax


def dynamic_stop(S_best, t, Smax, tmax):
    """dalle_bert.py:701-707 -> (Smax, tmax, took_new_best, stop)."""
    took = S_best > Smax
    if took:
        Smax, tmax = S_best, t
    return Smax, tmax, took, (t - tmax >= 5)
