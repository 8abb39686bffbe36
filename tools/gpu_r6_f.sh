#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py tests/test_round3_gpu.py tests/test_round6_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv or vqgan or split or strict or wide or census or encoder or tokens or strip or groupnorm" 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -5
bash tools/gpu_ab_lib.sh base "python tools/conv_layer_profile.py 54 bf16 | grep -E 'm0 (128x128 128|64x64 128|32x32)|total conv'; python bench.py --steps 30 --no-cpu-baseline --no-exact 2>&1 | grep -o '\"ms_per_step\": [0-9.]*' | head -1" 2
