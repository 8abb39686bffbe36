#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== decode tests"; timeout 900 python -m pytest tests/test_models_gpu.py tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "decode or artv" 2>&1 | tail -6 | cut -c1-300
for b in 1 4 8 16; do timeout 600 python bench.py --config 5 --batch $b --steps 1 --warmup 1 2>gpurun_out/c5_$b.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 batch $b:', round(d['ms_per_step'],1),'ms per call', round(d['value']),'sampled tok/s', round(d['roofline']['ms_per_token_step']*1e3),'us per token step')" || tail -3 gpurun_out/c5_$b.err; done
