#!/usr/bin/env python3
"""Headline benchmark: video-tokens/sec of one full MMVID training step on MI355X.

Workload (BASELINE.json configs[1]): text_to_video, 8 frames of 128x128, 64 text tokens
(L = 579), full BERT (CLIP ViT-B/32-shaped tower, 12 layers) + full VQGAN (encoder runs inside the step: 16
frames per sample), bf16 MFMA compute with fp32 accumulation / fp32 master weights, losses 7*MSM + 0.5*REL +
0.5*VID (three transformer passes), backward, clip_grad_norm_(1.0) and Adam -- i.e. everything train.py:298-325
of the reference does per iteration.  Synthetic inputs, random-init weights, per-GPU batch 6 (= the recipe's 48/8).

  python bench.py --gpus N --steps K --warmup W        (N > 1 via `python -m torch.distributed.run ...`)

Prints ONE JSON line on rank 0: value = whole-job video tokens / second (512 per sample), a `roofline` object for
the dominant kernel (HIP-event timed inside the timed region) and, at N = 1, a `cpu_baseline` object (the fp32
CPU oracle timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

# A GPU box shows all of the node's hardware threads (256) but grants a cgroup quota of a few cores.  Thread pools
# sized by the former (OpenMP, OpenBLAS) overrun the quota the moment they wake up and the kernel then throttles the
# whole process for the rest of the period -- including the thread that launches GPU work (cpu.stat: throttled 12.7 s
# over three benchmark runs before this).  Must happen before numpy / torch are imported.
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, '4')

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TEXT_LEN, FRAMES, SIZE, TOK_PER_SAMPLE = 64, 8, 128, 512
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PROF_EVERY = 10  # every 10th timed step (at least the last one) runs eagerly with per-launch HIP events; the others are graph replays
CLASS_NAMES = ['gemm_bf16_kernel<A.B^T> (forward)', 'gemm_bf16_kernel<dX>', 'gemm_bf16_kernel<dW>',
               'conv_igemm_kernel (VQGAN)', 'attn_fwd_kernel', 'attn_bwd (dq+dkv)']


def build_model(device, layers=12):
    from mmvid_amd.dalle_bert import BERT
    from mmvid_amd.vae import VQGanVAE1024
    vae = VQGanVAE1024(None, SIZE)
    vae.image_size = SIZE
    model = BERT(dim=768, vae=vae, cvae=None, num_text_tokens=49408, text_seq_len=TEXT_LEN,
                 which_transformer='openai_clip_visual', num_visuals=0, num_targets=FRAMES, transformer_layers=layers)
    return model.to(device)


def synth_batch(B, device, gen):
    text = torch.randint(1, 49408, (B, TEXT_LEN), generator=gen)
    lens = torch.randint(8, TEXT_LEN + 1, (B, ), generator=gen)
    text[torch.arange(TEXT_LEN)[None, :] >= lens[:, None]] = 0  # padded tail
    frames = torch.rand(B, FRAMES, 3, SIZE, SIZE, generator=gen)
    return text.to(device), frames.to(device)


def train_step(model, trainer, text, frames):
    trainer.zero_grad()
    lm, lr, lv = model(text, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                       msm_strategy_prob=MSM_PROB, msm_bernoulli_prob=MSM_BERN, vid_strategy_prob=VID_PROB)
    loss = 7.0 * lm + 0.5 * lr + 0.5 * lv
    loss.backward()
    trainer.step()
    return loss


MSM_PROB, MSM_BERN, VID_PROB = np.array([0.7, 0.1, 0.1, 0.1]), [0.2, 0.2], np.array([0.25, 0.25, 0.25, 0.25])


def loss_fn(model):
    """The device part of the step as a capture-safe function of tensors (engine.GraphedStep): the host-side random
    choices -- masking strategies and the VID warp (dalle_bert.py:992-1029, 1094) -- arrive as inputs."""
    def fn(text, frames, mask1, nfm, warped):
        lm, lr, lv = model(text, target=frames, return_loss=True, rel=True, vid=True, rel_no_fully_masked=True,
                           _mask1=mask1, _not_fully_masked=nfm, _target_warp=warped)
        return 7.0 * lm + 0.5 * lr + 0.5 * lv
    return fn


def host_random_inputs(model, text, frames):
    """Same RNG call order as BERT.forward (mask strategies first, then the warp)."""
    from mmvid_amd.dalle_bert import warp
    mask1, nfm = model._msm_mask(text.shape[0], text.device, MSM_PROB, MSM_BERN, 0)
    warped = warp(frames.detach(), VID_PROB).to(text.device)
    return {'text': text, 'frames': frames, 'mask1': mask1, 'nfm': nfm, 'warped': warped}


def pmc_traffic(prefix):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_pmc_{fetch,write}_size.csv: `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate runs, kernel-trace only).
    Units are KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so it is doubled
    (MI355X_MICROARCH.md, HBM; cross-checked on adam_kernel: 2 x FETCH = 16.0 B, WRITE = 14.0 B per parameter)."""
    import csv
    tot, disp = 0.0, 0
    try:
        for name, mult in (('fetch', 2.0), ('write', 1.0)):
            d = 0
            with open(os.path.join(ROOT, 'profiles', f'r01_pmc_{name}_size.csv')) as fh:
                for r in csv.DictReader(fh):
                    if r['kernel'].startswith(prefix):
                        tot += mult * float(r['total']) * 1024.0
                        d += int(r['dispatches'])
            disp = d
        return tot / disp if disp else None
    except OSError:
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
    cfg = ob.Cfg(sd, TEXT_LEN, 0, FRAMES, SIZE)
    opt = torch.optim.Adam([sd[k] for k in train_keys], lr=1e-4)
    gen = torch.Generator().manual_seed(1)
    text, frames = synth_batch(B, 'cpu', gen)

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
    return {'value': B * TOK_PER_SAMPLE / t, 'unit': 'video-tokens/s', 'cores': cores, 'kind': 'port',
            'sample': f'full training step at batch {B} (config 2 shapes, fp32 torch-CPU oracle): {note} step(s), {t:.2f} s/step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=6, help='per-GPU batch (even; the recipe is 48 / 8 GPUs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='launch every step from Python instead of replaying the captured step graph')
    ap.add_argument('--layers', type=int, default=12, help=argparse.SUPPRESS)  # debugging only; 12 = the model
    args = ap.parse_args()

    # a GPU box exposes all 256 hardware threads but a cgroup quota of a few cores: keep torch's CPU pool small so that
    # incidental host ops never fan out over hundreds of spinning OpenMP threads (cpu_baseline() sets its own count)
    torch.set_num_threads(min(4, host_cores()))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    under_launcher = 'RANK' in os.environ  # torch.distributed.run: join the group even when it has one member
    if world > 1 or under_launcher:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from mmvid_amd import _lib
    from mmvid_amd.build import build
    from mmvid_amd.engine import FlatTrainer, GraphedStep, backward_order, broadcast_parameters
    if local == 0:
        build()  # a no-op when the in-tree library is current (it is built by __graft_entry__.build())
    if dist.is_initialized():
        dist.barrier()
    # seed_everything(seed + rank) as train.py:87; identical initial weights come from the rank-0 broadcast
    seed = 42 + rank
    random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
    model = build_model(device, args.layers)
    broadcast_parameters(model)
    model.train()
    trainer = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order)
    gen = torch.Generator().manual_seed(seed)
    B = args.batch
    text, frames = synth_batch(B, device, gen)

    # Single process: the step is replayed as ONE hipGraph (the host then only draws the random masks / warp and
    # copies them in); with torch.distributed the eager step keeps the overlapped bucketed all-reduce.
    use_graph = world == 1 and not args.eager
    graph_warm = min(2, args.warmup) if use_graph else 0
    for _ in range(args.warmup - graph_warm):
        train_step(model, trainer, text, frames)
    graphed = None
    if use_graph:
        graphed = GraphedStep(trainer, loss_fn(model), host_random_inputs(model, text, frames), warmup=graph_warm)

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    lib = _lib.load()
    fence()
    lib.mmvid_prof_begin(1)
    lib.mmvid_prof_enable(0)
    t0 = time.perf_counter()
    host_s = 0.0  # time the host spends issuing the steps (it runs ahead of the GPU; ~= dt means host-bound)
    for i in range(args.steps):
        th = time.perf_counter()
        timed = i % PROF_EVERY == PROF_EVERY - 1 or args.steps < PROF_EVERY and i == args.steps - 1
        if timed or graphed is None:  # per-launch HIP events need direct launches
            lib.mmvid_prof_enable(1 if timed else 0)
            loss = train_step(model, trainer, text, frames)
            lib.mmvid_prof_enable(0)
        else:
            loss = graphed(**host_random_inputs(model, text, frames))
        host_s += time.perf_counter() - th
    fence()
    dt = time.perf_counter() - t0
    nc = len(CLASS_NAMES)
    ms, cnt, fl, tot = (ctypes.c_double * nc)(), (ctypes.c_int64 * nc)(), (ctypes.c_double * nc)(), (ctypes.c_int64 * nc)()
    lib.mmvid_prof_end(ms, cnt, fl, tot, nc)
    if dist.is_initialized():
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    gs = (ctypes.c_int64 * 3)()
    lib.mmvid_graph_stats(gs)
    graph_stats = {'direct': int(gs[0]), 'captured': int(gs[1]), 'replayed': int(gs[2])}
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * B * TOK_PER_SAMPLE / (dt / args.steps)
        print(f'[bench] {ms_per_step:.2f} ms/step, {value:.0f} video-tokens/s on {world} GPU(s); host issue time '
              f'{host_s / args.steps * 1e3:.2f} ms/step, load average {os.getloadavg()[0]:.1f} on {host_cores()} usable cores',
              file=sys.stderr, flush=True)
        kernels = []
        n_timed_steps = max(1, sum(1 for i in range(args.steps) if i % PROF_EVERY == PROF_EVERY - 1 or args.steps < PROF_EVERY and i == args.steps - 1))
        for i in range(nc):
            if cnt[i]:
                kernels.append({'kernel': CLASS_NAMES[i], 'timed_launches': int(cnt[i]), 'avg_ms': ms[i] / cnt[i],
                                'ms_per_step': ms[i] / n_timed_steps, 'launches_per_step': cnt[i] / n_timed_steps,
                                'tflops': fl[i] / (ms[i] * 1e-3) / 1e12})
        dom = max(kernels, key=lambda k: k['ms_per_step']) if kernels else None
        roofline = None
        if dom:
            roofline = {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': dom['tflops'], 'peak': PEAK_BF16_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': dom['tflops'] / PEAK_BF16_TFLOPS,
                        'traffic': pmc_traffic(dom['kernel'].split('<')[0].split(' ')[0]),
                        'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC passes committed under profiles/)',
                        'avg_launch_ms': dom['avg_ms'], 'launches_per_step': dom['launches_per_step'],
                        'timed_steps': n_timed_steps}
        out = {
            'metric': 'video-tokens/sec training step, 8-frame 128px text-to-video', 'value': value,
            'unit': 'video-tokens/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'text_to_video 8-frame 128x128, 64 text tokens (L=579), full dalle_bert (12-layer '
                                   'CLIP ViT-B/32 tower) + VQGAN encode in-step, MSM+REL+VID, backward, clip+Adam',
                       'per_gpu_batch': B, 'global_batch': world * B, 'seq_len': 579, 'parallelism': f'dp{world}', 'step_launch': 'hipGraph replay' if graphed is not None else 'eager',
                       'layers': args.layers},
            'loss': float(loss.detach()), 'roofline': roofline, 'kernels': kernels, 'graphs': graph_stats,
            'host_issue_ms_per_step': host_s / args.steps * 1e3, 'host_load_average': os.getloadavg()[0],
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(model)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
