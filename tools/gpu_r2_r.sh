#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== tests"; timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_models_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "positional or bert or trainer or graph or config" 2>&1 | tail -8 | cut -c1-300
echo "== launches"; timeout 300 python tools/step_ops.py 2>&1 | grep -v -E "amdgpu.ids|Warning|warn" | tail -30 | cut -c1-170
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 3 2>&1 >/dev/null | grep "bench\]" | cut -c1-120
