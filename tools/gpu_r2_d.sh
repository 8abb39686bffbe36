#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== decode step bench"; timeout 600 python tools/bench_decode_step.py 4 > gpurun_out/decode_step.log 2>&1; cut -c1-200 gpurun_out/decode_step.log | tail -25
echo "== captured exchange test"; timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k captured -s > gpurun_out/sel.log 2>&1; echo "rc=$?"; grep -E "captured exchange|passed|failed" gpurun_out/sel.log | cut -c1-250
echo "== bench under launcher, forced exchange"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --force-exchange > gpurun_out/bench_ddp1.log 2> gpurun_out/bench_ddp1.err; echo "rc=$?"; grep "bench\]" gpurun_out/bench_ddp1.err | cut -c1-250
grep -n "what()\|Error" gpurun_out/bench_ddp1.err | head -5 | cut -c1-250
python - <<'PY'
import json
for f in ('bench_ddp1',):
    try:
        d=json.loads(open(f'gpurun_out/{f}.log').read().strip().splitlines()[-1])
        print(f, 'ms/step',round(d['ms_per_step'],3),'value',round(d['value']),d['config'].get('step_launch',''))
        if 'gradient_exchange' in d: print('   exchange', d['gradient_exchange'])
    except Exception as e: print(f,'parse',e)
PY
