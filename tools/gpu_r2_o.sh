#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== conv tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "conv" 2>&1 | tail -6 | cut -c1-250
echo "== vqgan model tests"; timeout 900 python -m pytest tests/test_models_gpu.py tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "vqgan or strict or encoder or vid_negative or roundtrip" 2>&1 | tail -6 | cut -c1-250
echo "== per-op profile"; timeout 300 python tools/conv_layer_profile.py 54 2>&1 | grep -E "8x8|^total" | cut -c1-140
for sk in 0 1; do echo "== bench conv split-K $sk"; MMVID_CONV_SPLITK=$sk timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 3 2>&1 >/dev/null | grep "bench\]" | cut -c1-120; done
