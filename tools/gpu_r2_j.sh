#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== nan debug B=6"; timeout 300 python tools/debug_nan.py 6 6 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "== nan debug B=2"; timeout 300 python tools/debug_nan.py 2 4 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "== LN tests"; timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "layernorm" 2>&1 | tail -2 | cut -c1-200
echo "== strip microbench"; timeout 300 python tools/bench_gemm.py strip 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "== strip tests, schedule 2"
MMVID_STRIP_SCHED=2 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "strip or fused_groupnorm" 2>&1 | tail -2 | cut -c1-200
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/bench_j.log 2> gpurun_out/bench_j.err; grep "bench\]" gpurun_out/bench_j.err | cut -c1-220; python -c "
import json; d=json.loads(open('gpurun_out/bench_j.log').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], 'loss', d['loss'])"
