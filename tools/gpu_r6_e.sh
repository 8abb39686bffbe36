#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv or vqgan or split or strict or wide or census or encoder or tokens" 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -8
for m in bf16 split; do timeout 300 python tools/conv_layer_profile.py 54 $m 2>&1 | grep -v amdgpu > $o/conv_layers_54_$m.log; grep "8->128\|total conv" $o/conv_layers_54_$m.log; done
