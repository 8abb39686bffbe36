#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
python oracle/build.py >> gpurun_out/build.log 2>&1
echo "== selected tests"
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider \
  -k "gemv or layernorm or artv or captured or tower or graph or flat_trainer or config5" > gpurun_out/sel.log 2>&1; echo "sel rc=$?"
tail -4 gpurun_out/sel.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/sel.log | cut -c1-250
grep -E "captured exchange|fused decode|config 5" gpurun_out/sel.log | cut -c1-200
echo "== bench config 2"; timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; grep "bench\]" gpurun_out/bench.err | cut -c1-250
echo "== bench under launcher, forced exchange"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --force-exchange > gpurun_out/bench_ddp1.log 2> gpurun_out/bench_ddp1.err; echo "rc=$?"; grep "bench\]" gpurun_out/bench_ddp1.err | cut -c1-250
echo "== bench config 4"; timeout 600 python bench.py --config 4 --steps 10 --warmup 3 > gpurun_out/bench_c4.log 2> gpurun_out/bench_c4.err; echo "rc=$?"; grep "bench\]" gpurun_out/bench_c4.err | cut -c1-250
echo "== bench config 5"; timeout 900 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/bench_c5.log 2> gpurun_out/bench_c5.err; echo "rc=$?"; tail -3 gpurun_out/bench_c5.err | cut -c1-250
python - <<'PY'
import json
for f in ('bench','bench_ddp1','bench_c4','bench_c5'):
    try:
        d=json.loads(open(f'gpurun_out/{f}.log').read().strip().splitlines()[-1])
        print(f, 'ms/step',round(d['ms_per_step'],3),'value',round(d['value']),d['config'].get('step_launch',''), 'roof', d['roofline'] and {k:d['roofline'][k] for k in ('kernel','achieved','frac')})
        if 'gradient_exchange' in d: print('   exchange', d['gradient_exchange'])
        if 'artv_train_step' in d: print('   artv train', d['artv_train_step'])
        for k in d.get('kernels',[]): print('    ',k['kernel'],round(k['ms_per_step'],3),'ms',round(k['tflops'],1),'TF')
    except Exception as e: print(f,'parse',e)
PY
