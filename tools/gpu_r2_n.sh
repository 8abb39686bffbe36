#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== gemm tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "gemm" 2>&1 | tail -12 | cut -c1-250
echo "== gemm anatomy"; timeout 300 python tools/bench_gemm.py anatomy 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_anatomy.log | cut -c1-330
echo "== gemm microbench"; timeout 300 python tools/bench_gemm.py 2>&1 | grep -E "fwd|dX" | cut -c1-200
for w in 0 1; do echo "== bench wshape $w"; MMVID_GEMM_WSHAPE=$w timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 3 2>&1 >/dev/null | grep "bench\]" | cut -c1-120; done
