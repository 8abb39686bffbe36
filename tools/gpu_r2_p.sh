#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== vq tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "vq or vqgan or strict or roundtrip or tokens" 2>&1 | tail -4 | cut -c1-250
echo "== hbm rows microbench"; timeout 300 python tools/bench_hbm_rows.py 2>&1 | grep -v amdgpu.ids | grep -i -E "argmin|vq" | head -8 | cut -c1-200
echo "== per-op profile"; timeout 300 python tools/conv_layer_profile.py 54 2>&1 | grep -E "^vq|^total" | cut -c1-140
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 3 2>&1 >/dev/null | grep "bench\]" | cut -c1-120
echo "== artv b8"; timeout 600 python bench.py --config 5 --batch 8 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 b8', round(d['ms_per_step'],1),'ms', round(d['value']),'tok/s')"
