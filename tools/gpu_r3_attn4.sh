#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "attention or attn or tower or resident" 2>&1 | tail -3 > gpurun_out/attn4_tests.log
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn4_bench.log
timeout 300 python tools/attn_timeline.py 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/attn4_bench.log
timeout 300 python bench.py --steps 40 --warmup 5 --no-exact --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step']); print([ (k['kernel'][:20], round(k['ms_per_step'],3)) for k in d['kernels']])" >> gpurun_out/attn4_bench.log
cat gpurun_out/attn4_tests.log gpurun_out/attn4_bench.log
