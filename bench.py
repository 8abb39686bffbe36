#!/usr/bin/env python3
"""Headline benchmark: video-tokens/sec of one full MMVID training step on MI355X.

Default workload (--config 2 = BASELINE.json configs[1]): text_to_video, 8 frames of 128x128, 64 text
tokens (L = 579), full BERT (CLIP ViT-B/32-shaped tower, 12 layers) + full VQGAN (the encoder runs inside the step: 16
frames per sample), bf16 MFMA compute with fp32 accumulation / fp32 master weights, losses 7*MSM + 0.5*REL + 0.5*VID
(three transformer passes), backward, clip_grad_norm_(1.0) and Adam under the WarmupLR schedule -- everything
train.py:298-325 of the reference does per iteration, including the stochastic front-end (masking strategies, VID warp),
which is drawn on the device.  Synthetic inputs, random-init weights, per-GPU batch 6 (= the recipe's 48 / 8).
Other BASELINE configs, for driver-visible numbers next to the headline:
  --config 4   text_and_mask (one visual control frame through the cvae, vc_mode mask_8x8, L = 643), per-GPU batch 2
  --config 5   dalle_artv generate_images, 16 frames (1,024 sampled tokens per video, L = 1152), batch 4: a "step" is one
               call; value = sampled video tokens / s.  Also reports the ART-V training step.

  python bench.py --gpus N --steps K --warmup W        (N > 1 via `python -m torch.distributed.run ...`)

Prints ONE JSON line on rank 0: value = whole-job video tokens / second, a `roofline` object for the dominant kernel
(HIP events around every launch of one eagerly launched step right after the timed region) and, at N = 1 / config 2, a `cpu_baseline` object (the fp32 CPU oracle timed on
the host cores on a bounded sample of the same workload).  The step is ONE hipGraph replay at every N: with
torch.distributed the bucketed RCCL all-reduces are captured inside it, overlapped with the backward.
"""
import argparse
import ctypes
import json
import math
import os
import random
import sys
import time

# A GPU box shows all of the node's hardware threads (256) but grants a cgroup quota of a few cores.  Thread pools
# sized by the former (OpenMP, OpenBLAS) overrun the quota the moment they wake up and the kernel then throttles the
# whole process for the rest of the period -- including the thread that launches GPU work.  Must happen before numpy /
# torch are imported.
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, '4')

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TEXT_LEN, SIZE = 64, 128
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PROF_STEPS = 1  # eagerly launched steps with per-launch HIP events, run right AFTER the timed region (kernel-family table / roofline)
CLASS_NAMES = ['gemm_bf16_kernel<A.B^T> (forward)', 'gemm_bf16_kernel<dX>', 'gemm_bf16_kernel<dW>',
               'conv_igemm_kernel (VQGAN)', 'attn_fwd_kernel', 'attn_bwd (dq+dkv)']
MSM_PROB, MSM_BERN, VID_PROB = [0.7, 0.1, 0.1, 0.1], [0.2, 0.2], [0.25, 0.25, 0.25, 0.25]
WORKLOADS = {
    2: ('text_to_video 8-frame 128x128, 64 text tokens (L=579), full dalle_bert (12-layer CLIP ViT-B/32 tower) + VQGAN '
        'encode in-step, device front-end, MSM+REL+VID, backward, clip+Adam (WarmupLR)'),
    4: ('text_and_mask 8-frame 128x128 + 1 visual control frame through the cvae (vc_mode mask_8x8), 64 text tokens (L=643), '
        'full dalle_bert + 2 VQGANs in-step, MSM+REL+VID, backward, clip+Adam'),
    5: ('dalle_artv generate_images 16-frame 128x128 (1,024 sampled tokens per video over a KV cache, L=1152, 51,584 '
        'classes) + VQGAN decode'),
}


def build_model(cfg, device, layers=12):
    from mmvid_amd.vae import VQGanVAE1024
    vae = VQGanVAE1024(None, SIZE)
    vae.image_size = SIZE
    if cfg == 5:
        from mmvid_amd.dalle_artv import DALLE
        m = DALLE(dim=768, vae=vae, cvae=None, num_text_tokens=49408, text_seq_len=TEXT_LEN, which_transformer='openai_clip_visual',
                  num_visuals=1, num_targets=16, transformer_layers=layers)
        return m.to(device)
    from mmvid_amd.dalle_bert import BERT
    cvae = None
    if cfg == 4:
        cvae = VQGanVAE1024(None, SIZE)
        cvae.image_size = SIZE
    model = BERT(dim=768, vae=vae, cvae=cvae, num_text_tokens=49408, text_seq_len=TEXT_LEN,
                 which_transformer='openai_clip_visual', num_visuals=1 if cfg == 4 else 0, num_targets=8, transformer_layers=layers)
    return model.to(device)


def synth_batch(B, frames, device, gen, visuals=0):
    text = torch.randint(1, 49408, (B, TEXT_LEN), generator=gen)
    lens = torch.randint(8, TEXT_LEN + 1, (B, ), generator=gen)
    text[torch.arange(TEXT_LEN)[None, :] >= lens[:, None]] = 0  # padded tail
    out = {'text': text.to(device), 'frames': torch.rand(B, frames, 3, SIZE, SIZE, generator=gen).to(device)}
    if visuals:
        out['visual'] = torch.rand(B, visuals, 3, SIZE, SIZE, generator=gen).to(device)
    return out


def loss_fn(model, cfg):
    """The step's forward as a capture-safe function of the data tensors: every random choice is drawn on the device."""
    from mmvid_amd.functional import weighted_loss
    if cfg == 4:
        def fn(text, frames, visual):
            lm, lr, lv = model(text, visual=visual, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                               vc_mode='mask_8x8', msm_strategy_prob=MSM_PROB, msm_bernoulli_prob=MSM_BERN, vid_strategy_prob=VID_PROB)
            return weighted_loss((lm, lr, lv), (7.0, 0.5, 0.5))
        return fn

    def fn(text, frames):
        lm, lr, lv = model(text, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                           msm_strategy_prob=MSM_PROB, msm_bernoulli_prob=MSM_BERN, vid_strategy_prob=VID_PROB)
        return weighted_loss((lm, lr, lv), (7.0, 0.5, 0.5))  # train.py:320, one launch each way
    return fn


def eager_step(trainer, fn, batch):
    trainer.zero_grad()
    loss = fn(**batch)
    loss.backward()
    trainer.step()
    return loss.detach()


PMC_KERNELS = {  # bench kernel family -> kernel-name prefixes in the rocprofv3 PMC summaries
    'gemm_bf16_kernel<A.B^T> (forward)': ('gemm_bf16_kernel<false, false', 'gemm_bf16_lw_kernel<false, false'),
    'gemm_bf16_kernel<dX>': ('gemm_bf16_kernel<false, true', 'gemm_bf16_lw_kernel<false, true'),
    'gemm_bf16_kernel<dW>': ('gemm_bf16_kernel<true, true', 'gemm_bf16_lw_kernel<true, true', 'gemm_bf16_lw_grouped_kernel'),
    'conv_igemm_kernel (VQGAN)': ('conv_igemm_kernel', 'conv_strip_kernel', 'conv_in_kernel'),
    'attn_fwd_kernel': ('attn_fwd_kernel', ),
    'attn_bwd (dq+dkv)': ('attn_bwd_', ),
}


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc_{fetch,write}_size.csv: `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate runs, kernel-trace only;
    the newest round present is used).  Units are KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
    reads, so it is doubled (MI355X_MICROARCH.md, HBM; cross-checked on adam_kernel: 2 x FETCH = 16.0 B, WRITE = 14.0 B per
    parameter).  Bytes and dispatches are summed over every kernel of the family (both convolution kernels; one template
    instance per GEMM layout)."""
    import csv
    prefixes = PMC_KERNELS.get(family, (family.split('<')[0].split(' ')[0], ))
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        tot, disp = 0.0, {}
        try:
            for name, mult in (('fetch', 2.0), ('write', 1.0)):
                with open(os.path.join(ROOT, 'profiles', f'{rnd}_pmc_{name}_size.csv')) as fh:
                    for r in csv.DictReader(fh):
                        if r['kernel'].startswith(prefixes):
                            tot += mult * float(r['total']) * 1024.0
                            disp[name] = disp.get(name, 0) + int(r['dispatches'])
            if disp.get('fetch'):
                return tot / disp['fetch']
        except OSError:
            continue
    return None


def matrix_pipe_sustained_tflops(dev, seconds=0.7):
    """TFLOP/s of register-only bf16 MFMA chains on random operand bits, sustained for `seconds` (mmvid_probe 5): the ceiling the
    power-managed chip grants the matrix pipe, next to which `roofline.frac` (against the 2.5 PFLOP/s data-sheet peak) is to be read."""
    from mmvid_amd import _lib
    try:
        sink = torch.zeros(16, device=dev)
        arr = (ctypes.c_int32 * 3)(2000, 512, 1)
        flops = 2.0 * 32 * 32 * 16 * 32 * 2000 * 8 * 512
        call = lambda: _lib.call('mmvid_probe', 5, arr, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(4, int(seconds / 4.8e-3))
        a.record()
        for _ in range(n):
            call()
        b.record()
        torch.cuda.synchronize()
        return flops * n / (a.elapsed_time(b) * 1e-3) / 1e12
    except Exception as e:  # measurement only
        print(f'[bench] matrix-pipe ceiling probe failed: {e}', file=sys.stderr)
        return None


def host_cores():
    """Cores this process may actually use (affinity mask and cgroup quota, not the machine's core count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(model, B=2, budget_s=60.0):
    """The fp32 CPU oracle (oracle/bert.py, parity-pinned against the reference) on the host cores: the same step
    (3 passes fwd+bwd, 16 VQGAN encodes per sample, clip + Adam) at batch B -> video tokens / s.  Checker code,
    timed here only as the reported baseline.  Bounded sample: one warm-up step, then timed steps until
    `budget_s` is used (at least one, unless the warm-up alone blew the budget -- then the warm-up is reported)."""
    from oracle import bert as ob
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    train_keys = [k for k in sd if not k.startswith(('vae.', 'cvae.'))]
    for k in train_keys:
        sd[k].requires_grad_(True)
    cfg = ob.Cfg(sd, TEXT_LEN, 0, 8, SIZE)
    opt = torch.optim.Adam([sd[k] for k in train_keys], lr=1e-4)
    gen = torch.Generator().manual_seed(1)
    batch = synth_batch(B, 8, 'cpu', gen)
    text, frames = batch['text'], batch['frames']

    def step():
        t0 = time.time()
        with torch.no_grad():
            tt = ob.get_image_tokens(sd, cfg, frames)
            wt = ob.get_image_tokens(sd, cfg, frames.flip(1))  # VID negative: second encode of 8 frames
        mask1 = torch.rand(B, cfg.target_seq_len, generator=gen) < 0.2
        r = ob.forward_losses(sd, cfg, text, tt, mask1, wt)
        opt.zero_grad()
        (7 * r['loss_msm'] + 0.5 * r['loss_rel'] + 0.5 * r['loss_vid']).backward()
        torch.nn.utils.clip_grad_norm_([sd[k] for k in train_keys], 1.0)
        opt.step()
        return time.time() - t0

    warm = step()
    print(f'[bench] cpu baseline warm-up step {warm:.1f}s on {cores} threads', file=sys.stderr, flush=True)
    times, used = [], warm
    while used < budget_s and len(times) < 3:
        times.append(step())
        used += times[-1]
    t = float(np.mean(times)) if times else warm
    note = f'{len(times)} timed' if times else 'warm-up only (budget exceeded)'
    return {'value': B * 512 / t, 'unit': 'video-tokens/s', 'cores': cores, 'kind': 'port',
            'sample': f'full training step at batch {B} (config 2 shapes, fp32 torch-CPU oracle): {note} step(s), {t:.2f} s/step'}


def fence():
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def comm_report(trainer, step_ms, world):
    """Gradient exchange on its own (whole flat buffer, same bucketing), for the bus-bandwidth / overlap figures."""
    if (world == 1 and not trainer.force_exchange) or not dist.is_initialized():
        return None
    try:
        fence()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            trainer._send(0, trainer.numel)
            for w in trainer._works:
                w.wait()
            trainer._works = []
            torch.cuda.current_stream().wait_stream(trainer._comm_stream)
        fence()
        ar_ms = (time.perf_counter() - t0) / reps * 1e3
        dense = trainer.numel - sum(b - a for a, b in trainer._sparse_ranges())  # elements that go through the all-reduce
        wire = 2.0 * (world - 1) / world * 4.0 * dense  # bytes per GPU on the links, ring-equivalent
        return {'allreduce_alone_ms': ar_ms, 'bytes_per_gpu_on_wire': wire if world > 1 else 0.0,
                'bus_bandwidth_GBps': wire / (ar_ms * 1e-3) / 1e9 if world > 1 else None,  # a group of one is a loop-back: no links
                'gradient_bytes': 4.0 * trainer.numel, 'dense_allreduce_bytes': 4.0 * dense,
                'row_wise_tables': [n for n in getattr(trainer.model, 'sparse_grad_rows', dict)()],
                'note': 'overlap = 1 - (step - step_without_exchange) / allreduce_alone; the tables listed under row_wise_tables '
                        'are exchanged as (row id, row) pairs of the rows the batch touched, not all-reduced'}
    except Exception as e:  # never lose the headline line to a diagnostics failure
        return {'error': repr(e)}


def run_artv_sampling(args, device, rank, world):
    """--config 5: a 'step' = one DALLE.generate_images call (b videos x 1,024 sampled tokens + VQGAN decode)."""
    model = build_model(5, device, args.layers).eval()
    b = args.batch or 4
    gen = torch.Generator().manual_seed(42 + rank)
    batch = synth_batch(b, 1, device, gen, visuals=1)
    vis_tok = torch.randint(0, 1024, (b, 64), generator=gen).to(device)
    call = lambda: model.generate_images(batch['text'], visual=vis_tok)
    for _ in range(max(1, args.warmup)):
        call()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        images = call()[0]
    fence()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    # ART-V training step (dalle_artv.py:418-542) on the same model: tokens in, loss, backward, Adam
    from mmvid_amd.engine import FlatTrainer, backward_order
    model.train()
    tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order)
    tt = torch.randint(0, 1024, (b, 1024), generator=gen).to(device)
    def tstep():
        tr.zero_grad()
        loss = model(batch['text'], visual=vis_tok, target=tt, return_loss=True)[0]
        loss.backward()
        tr.step()
        return loss
    tstep(), tstep()
    fence()
    t1 = time.perf_counter()
    for _ in range(3):
        loss = tstep()
    fence()
    train_ms = (time.perf_counter() - t1) / 3 * 1e3
    if rank != 0:
        return
    per_call = dt / args.steps
    tokens = world * b * 1024
    # which form of the per-token tower step the decode session picks at this batch (mmvid_amd/clip_tower.py::DecodeSession)
    from mmvid_amd import _lib as _l
    import ctypes as _ct
    _cfg = model.transformer._cfg(b, 1152)
    if os.environ.get('MMVID_DECODE_PERSISTENT', '1') != '0' and _l.load().mmvid_tower_decode_persistent_supported(_ct.byref(_cfg), 1152):
        step_form = 'one persistent launch (256 co-resident blocks, tagged-word hand-over)'
    elif b <= 2:
        step_form = 'five launches per layer (vector-ALU matrix-vector kernels)'
    elif b <= 64:
        step_form = ('five launches per layer: linear layers on the matrix pipe (%d row block%s of v_mfma_f32_16x16x32_bf16 per wave, weights '
                     'streamed once), cached attention' % ((b + 15) // 16 if b > 16 else 1, 's' if b > 16 else ''))
    else:
        step_form = 'five launches per layer (matrix-pipe linear layers), slices of 64 sequences'
    # a decode step streams every tower weight once -- 12 layers x 7.08 M matrix params x 2 B (bf16) + the image block of the head -- and,
    # per SEQUENCE, the keys and values cached so far: at the loop's mean position (the 129-token prompt + half of the 1,024 sampled
    # tokens) layers x 2 (K, V) x 768 x 2 B each.  (Round 4 priced the batch-4 / batch-16 lines against the weights alone.)
    weight_bytes = args.layers * (4 * 768 * 768 + 2 * 768 * 3072) * 2 + 1024 * 768 * 2
    mean_pos = 129 + 1024 // 2
    cache_bytes = b * mean_pos * args.layers * 2 * 768 * 2
    stream_bytes = weight_bytes + cache_bytes
    step_s = per_call / 1024
    out = {'metric': 'sampled video-tokens/sec, dalle_artv generate_images, 16-frame 128px (BASELINE config 5; NOT the training-step metric)',
           'value': tokens / per_call, 'unit': 'sampled video-tokens/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': per_call * 1e3, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': WORKLOADS[5], 'config_id': 5, 'per_gpu_batch': b, 'seq_len': 1152, 'parallelism': f'replicas{world}',
                      'layers': args.layers, 'note': 'value counts SAMPLED tokens (inference), not training tokens'},
           'roofline': {'bound': 'hbm', 'kernel': 'decode step (weight streaming, batch %d)' % b, 'achieved': stream_bytes / step_s / 1e9,
                        'peak': 8000.0, 'unit': 'GB/s', 'frac': stream_bytes / step_s / 1e9 / 8000.0, 'traffic': None,
                        'ms_per_token_step': step_s * 1e3, 'algorithmic_bytes_per_step': stream_bytes, 'weight_bytes_per_step': weight_bytes,
                        'kv_cache_bytes_per_step_at_mean_position': cache_bytes, 'tower_step': step_form},
           'artv_train_step': {'ms_per_step': train_ms, 'video_tokens_per_s': b * 1024 / (train_ms * 1e-3), 'per_gpu_batch': b,
                               'loss': float(loss.detach())},
           'image_checksum': float(images.float().mean())}
    print(json.dumps(out))


MP_CONFIG = {'T1_n': 10, 'T2_n': 10, 'T3_n': 30, 'N1_n': 0.9, 'N2_n': 0.1, 'N3_n': 0.125, 'N4_n': 0.0625, 'T1_t': 10, 'T2_t': 5,
             'T3_t': 35, 'N1_t': 0., 'N2_t': 0., 'N3_t': 0., 'N4_t': 0., 'T': 20, 'B': 1}  # utils_args.py:221-281 defaults


def run_bert_sampling(args, device, rank, world):
    """--config 2|4 --sample: a 'step' = one BERT.generate_images call: control embedding, mask-predict with the reference's
    default schedule (mp_T = 20 tower passes per video, one candidate; scripts/*/test.sh), VQGAN decode of the 8 frames."""
    model = build_model(args.config, device, args.layers).eval()
    b = args.batch or 16  # scripts/mmvoxceleb/*/test.sh: --batch_size 16
    gen = torch.Generator().manual_seed(42 + rank)
    batch = synth_batch(b, 8, device, gen, visuals=1 if args.config == 4 else 0)
    kw = dict(visual=batch['visual']) if args.config == 4 else {}
    cfg = dict(MP_CONFIG, B=args.candidates)
    call = lambda: model.generate_images(batch['text'], mask_predict_steps=0, mp_config=cfg, dynamic=False, **kw)
    for _ in range(max(1, args.warmup)):
        call()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        images = call()[0]
    fence()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if rank != 0:
        return
    per_call = dt / args.steps
    L = model.total_seq_len
    passes = 1 + (cfg['T'] - 1) * cfg['B']  # sequences through the tower per video: step 0, then B candidates per step
    flops = world * b * passes * (12 * L * (24 * 768**2 + 4 * L * 768)) * args.layers / 12  # F(L) per pass and sample
    out = {'metric': 'generated video-tokens/sec, BERT.generate_images (mask-predict), 8-frame 128px (NOT the training-step metric)', 'value': world * b * 512 / per_call,
           'unit': 'video-tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': per_call * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': 'BERT.generate_images (mask-predict, mp_T=%d, %d candidate(s)) + VQGAN decode, 8 frames 128x128, '
                                  'config %d model' % (cfg['T'], cfg['B'], args.config), 'config_id': args.config,
                      'per_gpu_batch': b, 'seq_len': L, 'parallelism': f'replicas{world}', 'layers': args.layers,
                      'note': 'value counts GENERATED video tokens (inference), not training tokens'},
           'roofline': {'bound': 'mfma', 'kernel': 'tower forward passes of the sampler', 'achieved': flops / per_call / 1e12,
                        'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': flops / per_call / 1e12 / PEAK_BF16_TFLOPS, 'traffic': None,
                        'tower_passes_per_video': passes, 'ms_per_tower_pass_of_the_batch': per_call * 1e3 / cfg['T']},
           'image_checksum': float(images.float().mean())}
    print(f"[bench] BERT sampling: {per_call * 1e3:.1f} ms per generate_images call of {b} videos = {out['value']:.0f} generated "
          f"video-tokens/s", file=sys.stderr, flush=True)
    print(json.dumps(out))


def self_launch(n):
    """Re-run this command under `torch.distributed.run` with n ranks on this node (one process per GPU, rendezvous on
    127.0.0.1 at a free port).  stdout / stderr pass through, so rank 0's JSON line is this process's output."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'[bench] --gpus {n} without a launcher: starting {n} ranks ({" ".join(cmd[1:8])} ...)', file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def dry_run(args, world, rank):
    """Arg / rank plumbing without a GPU: join a gloo group, check every rank agrees on the arguments, print the skeleton."""
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1), float(args.steps), float(args.gpus)])
        dist.all_reduce(t)
        assert t[0].item() == world * (world + 1) / 2 and t[1].item() == world * args.steps and t[2].item() == world * args.gpus
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        B = args.batch or (6 if args.config == 2 else 2)
        print(json.dumps({'metric': 'video-tokens/sec training step, 8-frame 128px text-to-video', 'value': None, 'unit': 'video-tokens/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'dry_run': True, 'scaling': 'weak',
                          'config': {'workload': WORKLOADS[args.config], 'per_gpu_batch': B, 'global_batch': world * B,
                                     'parallelism': f'dp{world}'}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=2, choices=sorted(WORKLOADS), help='BASELINE.json config (1-based): 2 headline, 4, 5')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (even; config 2: the recipe is 48 / 8 GPUs)')
    ap.add_argument('--sample', action='store_true', help='config 2 / 4: time BERT.generate_images (mask-predict) instead of training')
    ap.add_argument('--candidates', type=int, default=1, help='--sample: beam candidates per step (mp_B)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='launch every step from Python instead of replaying the captured step graph')
    ap.add_argument('--force-exchange', action='store_true', help='run the gradient all-reduce path even with one rank (tests)')
    ap.add_argument('--layers', type=int, default=12, help=argparse.SUPPRESS)  # debugging only; 12 = the model
    ap.add_argument('--dry-run', action='store_true', help='rank plumbing only (gloo, no GPU work): start the ranks, all-reduce '
                    'one scalar, print the JSON skeleton')
    ap.add_argument('--no-exact', action='store_true', help="skip the extra legs that time the step with vae.strict = 'split' (index-exact) and 'mixed' (99.9 %)")
    ap.add_argument('--strict', nargs='?', const='fp32', default=None, choices=['fp32', 'split', 'mixed'],
                    help="run the VQGAN encoder in an exact-index mode: 'fp32' (vae.strict = True, fp32 matrix pipe) or 'split' "
                    "(vae.strict = 'split': bf16-pair convolutions, 3 products each, on the bf16 pipe)")
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher (train.py:47-66 spawns its ranks from main() the
        # same way); the ranks run this file again with RANK / LOCAL_RANK / WORLD_SIZE set and rank 0 prints the JSON line
        raise SystemExit(self_launch(args.gpus))
    if args.steps is None:
        args.steps = 2 if args.config == 5 else (3 if args.sample else 50)  # 50 graph replays = ~0.85 s timed

    # a GPU box exposes all 256 hardware threads but a cgroup quota of a few cores: keep torch's CPU pool small so that
    # incidental host ops never fan out over hundreds of spinning OpenMP threads (cpu_baseline() sets its own count)
    torch.set_num_threads(min(4, host_cores()))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks')
    if args.dry_run:
        return dry_run(args, world, rank)
    ndev = torch.cuda.device_count()
    # MMVID_BENCH_SHARED_GPU=1 (a check of the world > 1 GPU path on a 1-GPU box, never a measurement): the ranks share the devices that
    # exist and exchange through gloo -- RCCL refuses two ranks on one device; gloo collectives cannot be captured, so the step runs eagerly
    shared = os.environ.get('MMVID_BENCH_SHARED_GPU', '0') == '1' and ndev >= 1
    if (ndev < world or local >= ndev) and not shared:
        raise SystemExit(f'[bench] --gpus {args.gpus}: only {ndev} device(s) visible on this node (rank {rank}, local rank {local}); '
                         'one process per GPU needs as many devices as ranks')
    dev_index = local % ndev if shared else local
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    under_launcher = 'RANK' in os.environ  # torch.distributed.run: join the group even when it has one member
    if world > 1 or under_launcher:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if shared and ndev < world:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from mmvid_amd import _lib
    from mmvid_amd.build import build
    from mmvid_amd.engine import FlatTrainer, GraphedStep, WarmupLR, backward_order, broadcast_parameters
    if local == 0:
        build()  # a no-op when the in-tree library is current (it is built by __graft_entry__.build())
    if dist.is_initialized():
        dist.barrier()
    # seed_everything(seed + rank) as train.py:87; identical initial weights come from the rank-0 broadcast
    seed = 42 + rank
    random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
    if args.sample and args.config != 5:
        run_bert_sampling(args, device, rank, world)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if args.config == 5:
        run_artv_sampling(args, device, rank, world)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    model = build_model(args.config, device, args.layers)
    if args.strict:  # exact-index tokenisation (vae.strict; 'split' = 3-term bf16 split on the MFMA pipe, True = f32 MFMA)
        model.vae.strict = True if args.strict == 'fp32' else args.strict
        if model.cvae is not None:
            model.cvae.strict = model.vae.strict
    model.frontend.seed = seed  # every rank draws its own masks / warps
    broadcast_parameters(model)
    model.train()
    trainer = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order, force_exchange=args.force_exchange,
                          lr_schedule=WarmupLR(1e-6, 1e-4, 5000, every=1))
    gen = torch.Generator().manual_seed(seed)
    B = args.batch or (6 if args.config == 2 else 2)
    batch = synth_batch(B, 8, device, gen, visuals=1 if args.config == 4 else 0)
    fn = loss_fn(model, args.config)
    tok_per_sample = 512

    # The step is replayed as ONE hipGraph at every world size (with torch.distributed the bucketed all-reduces are captured
    # inside it, between the backward chunks they overlap with).  The host only calls replay.
    use_graph = not args.eager
    capture_note = None
    if use_graph and dist.is_initialized():
        from mmvid_amd.engine import collectives_capturable
        if not collectives_capturable(device):  # asked on a throw-away group: a collective that fails inside the step's capture would
            use_graph = False                   # take the eager fall-back down with it (mmvid_amd/engine.py)
            capture_note = 'eager (the %s backend cannot capture a collective)' % dist.get_backend()
    graph_warm = min(2, args.warmup) if use_graph else 0
    for _ in range(args.warmup - graph_warm):
        eager_step(trainer, fn, batch)
    graphed, step_launch = None, capture_note or 'eager'
    if use_graph:
        graphed = GraphedStep(trainer, fn, batch, warmup=graph_warm)
        step_launch = 'hipGraph replay' if graphed.graph is not None else f'eager (capture failed: {graphed.capture_error})'

    lib = _lib.load()
    fence()
    lib.mmvid_prof_begin(1)
    lib.mmvid_prof_enable(0)
    t0 = time.perf_counter()
    host_s = 0.0  # time the host spends issuing the steps (it runs ahead of the GPU; ~= dt would mean host-bound)
    for i in range(args.steps):  # the timed region: exactly `steps` steps as the product runs them (one graph replay each)
        th = time.perf_counter()
        if graphed is None or graphed.graph is None:
            loss = eager_step(trainer, fn, batch)
        else:
            loss = graphed()
        host_s += time.perf_counter() - th
    fence()
    dt = time.perf_counter() - t0
    # Per-kernel-family timing: HIP events around every launch need direct launches (they cannot be recorded inside a graph
    # replay), so PROF_STEPS extra steps of the SAME step run eagerly right after the timed region.  (They used to replace
    # every 10th timed step: an eagerly launched step is ~600 launches and 25-150 ms on a loaded host, which moved the
    # headline by 0.6-3 ms per step depending on the box.)
    for _ in range(PROF_STEPS):
        lib.mmvid_prof_enable(1)
        eager_step(trainer, fn, batch)
        lib.mmvid_prof_enable(0)
    fence()
    nc = len(CLASS_NAMES)
    ms, cnt, fl, tot = (ctypes.c_double * nc)(), (ctypes.c_int64 * nc)(), (ctypes.c_double * nc)(), (ctypes.c_int64 * nc)()
    lib.mmvid_prof_end(ms, cnt, fl, tot, nc)
    if dist.is_initialized():
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    comm = comm_report(trainer, dt / args.steps * 1e3, world)
    if comm and 'error' not in comm and graphed is not None and graphed.graph is not None:
        try:  # the same captured step without the exchange -> how much of the all-reduce is hidden behind the backward
            trainer.exchange_enabled = False
            nocomm = GraphedStep(trainer, fn, batch, warmup=0)
            fence()
            t1 = time.perf_counter()
            for _ in range(5):
                nocomm()
            fence()
            base_ms = (time.perf_counter() - t1) / 5 * 1e3
            trainer.exchange_enabled = True
            exposed = max(0.0, dt / args.steps * 1e3 - base_ms)
            comm.update(step_without_exchange_ms=base_ms, exposed_ms=exposed,
                        overlap_frac=max(0.0, 1.0 - exposed / comm['allreduce_alone_ms']))
        except Exception as e:
            comm['overlap_error'] = repr(e)

    # The same step with the tokeniser in its more exact arithmetic modes.  Reported beside the headline, never as `value`.  Round 6 (the flip
    # census of 40,960 fresh tokens per mode, tools/flip_census.py -> profiles/r06_flip_census.log; 2 x 1,024 reference tokens in
    # tests/test_round6_gpu.py): the headline's bf16 tokeniser agrees with the reference on 97.7 % of the tokens; 'mixed' on 99.87 % --
    # NOT an exact mode, rounds 4-5 called it one on 64 + 384 golden tokens; 'split' on all but ties at the reference's own fp32
    # resolution (2 tokens in 40,960 against the fp32 mode, 1 in 2,048 golden tokens: a top-2 gap of 17 fp32 spacings of the distance);
    # the fp32 mode (vae.strict = True, `--strict fp32`) on every token seen.  `exact_index_step` is therefore the 'split' step.
    exact = None
    if world == 1 and not args.strict and not args.eager and not args.no_exact:
        what = {'mixed': "NOT index-exact: 99.87 % of the reference's tokens (flip census: 57 of 40,960 against the fp32 mode; 5 of 2,048 "
                         "golden tokens).  The pair operator of 'split' except the 3x3 residual-block convolutions of the 128x128, 64x64 and "
                         '32x32 levels (82 % of the multiply-adds), which are ONE product of fp16 operands, fp32 accumulate',
                'split': 'every VQGAN convolution as three bf16 products of hi/lo pairs (fp32 accumulate), fp32 GroupNorm / attention / residual '
                         "stream: token indices equal the reference's except ties at its own fp32 resolution (flip census: 2 of 40,960 against "
                         'the fp32 mode, 0 of 4,096 against the CPU oracle; 1 of 2,048 golden tokens, top-2 gap = 17 fp32 spacings of the distance)'}
        census = {'mixed': {'flips_vs_fp32_mode': 57, 'tokens': 40960, 'flips_vs_reference_goldens': 5, 'golden_tokens': 2048},
                  'split': {'flips_vs_fp32_mode': 2, 'tokens': 40960, 'flips_vs_reference_goldens': 1, 'golden_tokens': 2048}}
        try:
            for mode in ('split', 'mixed'):
                model.vae.strict = mode
                if model.cvae is not None:
                    model.cvae.strict = mode
                ex_step = GraphedStep(trainer, fn, batch, warmup=2)
                n_ex = max(5, min(20, args.steps))
                fence()
                t1 = time.perf_counter()
                for _ in range(n_ex):
                    ex_step()
                fence()
                ex_ms = (time.perf_counter() - t1) / n_ex * 1e3
                leg = {'vae.strict': mode, 'steps': n_ex, 'ms_per_step': ex_ms, 'value': B * tok_per_sample / (ex_ms * 1e-3),
                       'unit': 'video-tokens/s', 'ratio_to_headline_step': ex_ms / (dt / args.steps * 1e3),
                       'launch': 'hipGraph replay' if ex_step.graph is not None else 'eager', 'what': what[mode],
                       'flip_census': dict(census[mode], source='profiles/r06_flip_census.log, tests/test_round6_gpu.py')}
                if exact is None:
                    exact = leg
                else:
                    exact['near_exact_mixed_operator'] = leg
                del ex_step
        except Exception as e:
            exact = dict(exact or {}, error=repr(e))
        finally:
            model.vae.strict = False
            if model.cvae is not None:
                model.cvae.strict = False

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * B * tok_per_sample / (dt / args.steps)
        print(f'[bench] {ms_per_step:.2f} ms/step, {value:.0f} video-tokens/s on {world} GPU(s); host issue time '
              f'{host_s / args.steps * 1e3:.2f} ms/step, load average {os.getloadavg()[0]:.1f} on {host_cores()} usable cores',
              file=sys.stderr, flush=True)
        kernels = []
        n_timed_steps = PROF_STEPS
        for i in range(nc):
            if cnt[i]:
                kernels.append({'kernel': CLASS_NAMES[i], 'timed_launches': int(cnt[i]), 'avg_ms': ms[i] / cnt[i],
                                'ms_per_step': ms[i] / n_timed_steps, 'launches_per_step': cnt[i] / n_timed_steps,
                                'tflops': fl[i] / (ms[i] * 1e-3) / 1e12})
        dom = max(kernels, key=lambda k: k['ms_per_step']) if kernels else None
        roofline = None
        sustained = matrix_pipe_sustained_tflops(device) if rank == 0 else None
        if dom:
            roofline = {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': dom['tflops'], 'peak': PEAK_BF16_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': dom['tflops'] / PEAK_BF16_TFLOPS,
                        'traffic': pmc_traffic(dom['kernel']),
                        'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC passes committed under profiles/)',
                        'avg_launch_ms': dom['avg_ms'], 'launches_per_step': dom['launches_per_step'],
                        'timed_steps': n_timed_steps,
                        # what the matrix pipe sustains on THIS box with nothing but MFMAs in flight (register-only
                        # v_mfma_f32_32x32x16_bf16 chains on random operand bits, ~0.7 s; csrc/probe.hip): the chip clocks down to its
                        # power limit (1.8-1.9 GHz, ~1.8 PFLOP/s; 2.46 PFLOP/s on all-zero operands, profiles/r04_power_probe_*.log)
                        'sustained_mfma_ceiling': sustained, 'frac_of_sustained_ceiling': (dom['tflops'] / sustained) if sustained else None,
                        'measured_in': 'HIP events around every launch of %d eagerly launched step(s) of the same workload, run right '
                                       'after the timed region (events cannot be recorded inside a hipGraph replay)' % n_timed_steps}
        L = model.total_seq_len
        _lib.check_device_faults()  # an out-of-range token id / CE target anywhere in the run is an error, not a silent row 0
        loss_value = float(loss.detach())
        if not math.isfinite(loss_value):  # a step that produced NaN/Inf measured nothing: fail loudly instead of reporting it
            raise SystemExit(f'[bench] non-finite loss {loss_value} after {trainer.step_count} optimizer steps: the run is invalid')
        out = {
            'metric': 'video-tokens/sec training step, 8-frame 128px text-to-video', 'value': value,
            'unit': 'video-tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {None: 'bf16', 'fp32': 'bf16 (transformer) + fp32 (VQGAN encoder, exact token indices)',
                      'split': 'bf16 (transformer) + bf16-pair convolutions / fp32 elsewhere (VQGAN encoder)',
                      'mixed': 'bf16 (transformer) + fp16 / bf16-pair convolutions / fp32 elsewhere (VQGAN encoder)'}[args.strict],
            'data': 'synthetic',
            'config': {'workload': WORKLOADS[args.config] + {None: '', 'fp32': ' [vae.strict: fp32 encoder]',
                                                             'split': " [vae.strict = 'split': bf16-pair encoder]",
                                                             'mixed': " [vae.strict = 'mixed': fp16 + bf16-pair encoder]"}[args.strict],
                       'config_id': args.config, 'per_gpu_batch': B, 'global_batch': world * B,
                       'seq_len': L, 'parallelism': f'dp{world}' + (' (ranks SHARE the visible GPUs, gloo exchange: a path check, not a measurement)' if os.environ.get('MMVID_BENCH_SHARED_GPU', '0') == '1' and torch.cuda.device_count() < world else ''), 'step_launch': step_launch, 'layers': args.layers},
            'loss': loss_value, 'roofline': roofline, 'kernels': kernels,
            'host_issue_ms_per_step': host_s / args.steps * 1e3, 'host_load_average': os.getloadavg()[0],
            'lr_device_scalar': float(trainer._lr_dev), 'optimizer_steps': trainer.step_count,
        }
        if comm:
            out['gradient_exchange'] = comm
        if exact:
            out['exact_index_step'] = exact
        if world == 1 and args.config == 2 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(model)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
