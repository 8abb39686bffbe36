#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== gemm tests, persistent"; MMVID_GEMM_PERSIST=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "gemm or tower or bert_training or reproducible" 2>&1 | tail -3 | cut -c1-250
echo "== A/B"; timeout 600 python tools/ab_graph.py gemm_persist 0 1 2>&1 | grep median | cut -c1-60,120-200
