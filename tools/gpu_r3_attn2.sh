#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py -x -q -k "attention or attn" 2>&1 | tail -3 > gpurun_out/attn2_tests.log
for f in 0 1; do echo "attn_res=$f" >> gpurun_out/attn2_bench.log; MMVID_ATTN_RES=$f timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/attn2_bench.log; done
timeout 300 python bench.py --steps 40 --warmup 5 --no-exact --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step']); print([ (k['kernel'][:20], round(k['ms_per_step'],3)) for k in d['kernels']])" >> gpurun_out/attn2_bench.log
cat gpurun_out/attn2_tests.log gpurun_out/attn2_bench.log
