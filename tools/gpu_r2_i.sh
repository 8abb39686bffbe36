#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== strip microbench"; timeout 300 python tools/bench_gemm.py strip 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "== strip tests both schedules"
for sc in 0 1 2; do MMVID_STRIP_SCHED=$sc timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "strip or fused_groupnorm" 2>&1 | tail -2 | cut -c1-200; done
