"""Batched, device-resident mask-predict sampler for BERT (mmvid_pytorch/dalle_bert.py:514-714).

The reference samples one video at a time with batch-1 tower passes, `Tmax * B` of them per video, and reads a score
back to the host for every candidate.  Here every video of the call and every beam candidate of a step go through the
tower as ONE batch of `b * B` sequences; confidences, tokens, keep-masks, scores, the best-candidate choice and the
dynamic stop live on the device (csrc/sample.hip).  The only host read is one "is anyone still running" flag per step
when `dynamic` is set.

Results are a function of the Exp(1) race variates (ops.exponential_like in production; tests inject them), of the
schedule and of the model -- independent of how many videos share the call.
"""
import numpy as np
import torch

from . import ops


def schedule(mp_config, N):
    """Keep-count and noise schedules of dalle_bert.py:586-614: three linear / constant segments each."""
    c = mp_config
    seg_n = [N * np.linspace(c['N1_n'], c['N2_n'], c['T1_n']), np.full(c['T2_n'], max(1, int(N * c['N3_n']))),
             np.full(c['T3_n'], max(1, int(N * c['N4_n'])))]
    seg_t = [np.linspace(c['N1_t'], c['N2_t'], c['T1_t']), np.full(c['T2_t'], c['N3_t']), np.full(c['T3_t'], c['N4_t'])]
    n = [int(v) for v in np.concatenate(seg_n)]
    temp = [float(v) for v in np.concatenate(seg_t)]
    return n, temp


def preserved_tokens(model, b, preserve, t_overlap, long_mode, device):
    """Which target positions are given (dalle_bert.py:542-583) -> (N, keep_fixed uint8 [TS] | None, fixed_tok [b, TS]).
    'long': the last t_overlap frames of the previous clip become the first frames of this one;
    'interp*': the given T/2 frames fill the even frame slots."""
    TS, ISL, T = model.target_seq_len, model.image_seq_len, model.num_targets
    MASK = model.image_token_lut['[MASK]']
    fixed_tok = torch.full((b, TS), MASK, dtype=torch.long, device=device)
    frame_of = torch.arange(TS, device=device) // ISL
    interp = long_mode in ('interp', 'interp2', 'interp_real')
    if long_mode == 'long':
        if preserve is None:
            return TS, None, fixed_tok
        prev = preserve.reshape(b, T * preserve.shape[-1])
        fixed = frame_of < t_overlap
        fixed_tok[:, :ISL * t_overlap] = prev[:, TS - ISL * t_overlap:]
        return TS - ISL * t_overlap, fixed.to(torch.uint8), fixed_tok
    if interp:
        if preserve is None:
            return TS // 2, None, fixed_tok
        given = preserve.reshape(b, T, ISL)[:, :T // 2]
        fixed_tok.view(b, T, ISL)[:, ::2] = given
        return TS // 2, (frame_of % 2 == 0).to(torch.uint8), fixed_tok
    return TS, None, fixed_tok


@torch.no_grad()
def mask_predict(model, control_emb, dynamic=True, debug=False, steps=10, preserve=None, t_overlap=1, mp_config=None,
                 long_mode='long', race=None, trace=None):
    """-> (tokens [b, TS] int64, image_samples list).  `race(name, shape)` supplies the Exp(1) variates (default: the
    device generator); `trace` (a list) receives one dict of the step's tensors per step (tests)."""
    dev = control_emb.device
    b, csl, E = control_emb.shape
    TS, MASK, V = model.target_seq_len, model.image_token_lut['[MASK]'], model.num_image_tokens
    L = csl + TS
    draw = race if race is not None else (lambda name, shape: ops.exponential_like(shape, dev))
    N, fixed, fixed_tok = preserved_tokens(model, b, preserve, t_overlap, long_mode, dev)
    n, temp = schedule(mp_config, N)
    Tmax = mp_config['T'] if steps <= 0 else steps
    Bm = mp_config['B']
    if Tmax < 2:
        raise RuntimeError('mask_predict needs at least 2 steps (the reference returns nothing for steps == 1)')

    control_emb = ops._chk(control_emb.contiguous().float(), torch.float32, 'control_emb')
    iemb = model.image_emb.weight.detach()
    tpos = model.target_pos_emb.table().detach().contiguous()
    rel_head, vid_head = model.to_logits_rel, model.to_logits_vid

    def tower_logits(I_in, mask1, nb):
        x = ops.mp_build_input(control_emb, iemb, tpos, I_in, mask1, nb, MASK)
        out = model.transformer_forward(x)  # [b*nb, L, E]
        rows = out[:, csl:, :].reshape(b * nb * TS, E)
        return out, model.to_logits_rows(rows)

    def noise(name, t, shape):
        if temp[t] == 0.0:
            return None
        return (race(name + '_noise_u', shape) if race is not None else torch.rand(shape, device=dev))

    # ---- step 0: everything that is not given is [MASK]
    out, logits = tower_logits(fixed_tok, None, 1)
    E0 = draw('tok0', (b * TS, V))
    I_new, Y = ops.sample_race(logits, E0, noise('tok0', 0, (b * TS, V)), temp[0])
    Y = Y.view(b, TS)
    I_tok = I_new.view(b, TS)
    if fixed is not None:
        I_tok = torch.where(fixed.bool(), fixed_tok, I_tok)
    I_tok = I_tok.contiguous()
    if trace is not None:
        trace.append(dict(t=0, logits=logits, E_tok=E0, Y=Y.clone(), I_tok=I_tok.clone()))
    Imax = I_tok.clone()
    Smax = torch.zeros(b, device=dev)
    tmax = torch.zeros(b, dtype=torch.int32, device=dev)
    active = torch.ones(b, dtype=torch.uint8, device=dev)
    seq0 = torch.arange(b * Bm, device=dev) * L
    rel_rows, vid_rows = (seq0 + model.rel_tok_index).contiguous(), (seq0 + model.vid_tok_index).contiguous()
    frames = [[model.decode_images(I_tok[i:i + 1])] for i in range(b)] if debug else None
    stopped_at = [None] * b

    for t in range(1, Tmax):
        Ek = draw(f'keep{t}', (b, Bm, TS))
        mask1 = ops.mp_select_keep(Y, Ek, fixed, N - n[t - 1])
        out, logits = tower_logits(I_tok, mask1, Bm)
        Et = draw(f'tok{t}', (b * Bm * TS, V))
        Inew, Ynew = ops.sample_race(logits, Et, noise(f'tok{t}', t, (b * Bm * TS, V)), temp[t])
        out2d = out.view(b * Bm * L, E)
        z_rel = ops.head_rows_fwd(out2d, rel_rows, rel_head[0].weight, rel_head[0].bias, rel_head[1].weight.view(-1),
                                  rel_head[1].bias, rel_head[0].eps)[0]
        z_vid = ops.head_rows_fwd(out2d, vid_rows, vid_head[0].weight, vid_head[0].bias, vid_head[1].weight.view(-1),
                                  vid_head[1].bias, vid_head[0].eps)[0]
        rec = None
        if trace is not None or debug:
            rec = dict(t=t, k=N - n[t - 1], E_keep=Ek, mask1=mask1, logits=logits, E_tok=Et, Ynew=Ynew.view(b, Bm, TS),
                       Inew=Inew.view(b, Bm, TS), z_rel=z_rel, z_vid=z_vid, Y_before=Y.clone(), I_before=I_tok.clone(),
                       active_before=active.clone(), S=torch.empty(b, Bm, device=dev),
                       jmax=torch.empty(b, dtype=torch.int32, device=dev))
        ops.mp_update(mask1, Ynew, Inew, z_rel, z_vid, t, dynamic, Y, I_tok, Imax, Smax, tmax, active,
                      rec['S'] if rec else None, rec['jmax'] if rec else None)
        if rec is not None and trace is not None:
            rec.update(Y=Y.clone(), I_tok=I_tok.clone(), Imax=Imax.clone(), Smax=Smax.clone(), tmax=tmax.clone(),
                       active=active.clone())
            trace.append(rec)
        if debug:  # per video, in the reference's order: the masked previous sample, then the new one
            was, jm = rec['active_before'].cpu(), rec['jmax'].cpu()
            for i in range(b):
                if not was[i]:
                    continue
                hidden = (mask1[i, jm[i]] == 0).float().unsqueeze(0)
                frames[i].append(torch.clamp(frames[i][-1] * 0.7 + model.decode_masks(hidden) * 0.4, 0, 1))
                frames[i].append(model.decode_images(I_tok[i:i + 1]))
        if dynamic and not bool(active.any()):  # the one host read of a step
            break
    image_samples = [f for per_video in frames for f in per_video] if debug else []
    return Imax, image_samples
