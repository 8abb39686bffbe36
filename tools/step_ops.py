"""Which framework (non-library) launches does one config-2 training step still contain?  Runs a few eager steps under
torch.profiler and prints every aten operator that launched a device kernel, with input shapes and the Python call site.
python tools/step_ops.py"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmvid_amd.engine import FlatTrainer, WarmupLR, backward_order  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(42)
model = bench.build_model(2, dev, 12)
model.train()
tr = FlatTrainer(model, lr=1e-4, max_grad_norm=1.0, order=backward_order, lr_schedule=WarmupLR(1e-6, 1e-4, 5000, every=1))
batch = bench.synth_batch(6, 8, dev, torch.Generator().manual_seed(42))
fn = bench.loss_fn(model, 2)
for _ in range(3):
    bench.eager_step(tr, fn, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    bench.eager_step(tr, fn, batch)
    torch.cuda.synchronize()
rows = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.self_device_time_total > 0:
        site = ''
        for fr in (ev.stack or []):
            if '/mmvid_amd/' in fr or 'bench.py' in fr:
                site = fr.split('/')[-1][:60]
                break
        key = (ev.name, str(ev.input_shapes)[:70], site)
        rows[key][0] += 1
        rows[key][1] += ev.self_device_time_total
# device-to-device copies issued through the runtime (hipMemcpyAsync: blit kernels, not aten kernels)
mem = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU and ('emcpy' in ev.name or 'emset' in ev.name or 'copyBuffer' in ev.name):
        mem[ev.name][0] += 1
        mem[ev.name][1] += ev.device_time_total if hasattr(ev, 'device_time_total') else ev.cuda_time_total
for name, (n, us) in sorted(mem.items(), key=lambda kv: -kv[1][1]):
    print(f'{us:9.1f} us {n:4d}x  {name}')
calls = defaultdict(int)
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ('hipMemcpy' in ev.name or 'hipMemset' in ev.name):
        site = ''
        for fr in (ev.stack or []):
            if '/mmvid_amd/' in fr or 'bench.py' in fr:
                site = fr.split('/')[-1][:70]
                break
        calls[(ev.name, site)] += 1
for (name, site), n in sorted(calls.items(), key=lambda kv: -kv[1]):
    print(f'   runtime call {n:4d}x  {name:24s} {site}')
tot = 0.0
for (name, shp, site), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print(f'{us:9.1f} us {n:4d}x  {name:28s} {shp:70s} {site}')
print(f'total {tot / 1e3:.3f} ms in {sum(v[0] for v in rows.values())} framework launches')
