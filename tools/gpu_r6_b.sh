#!/bin/bash
# Round-6 session B: the MFMA decode gemv -- tests, then the decode step at batch 4 / 8 / 16 with the old (vector-ALU) and new forms.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
timeout 600 python -m pytest tests/test_round6_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemv or decode or artv or kv_cache" 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -5
for b in 4 8 16; do for m in 99 3; do echo "== batch $b MMVID_GEMV16_MIN=$m"; MMVID_GEMV16_MIN=$m timeout 300 python tools/bench_decode_step.py $b 2>&1 | grep -v -i "warn\|amdgpu.ids" | grep -E "gemv|position" ; done; done | tee $o/decode_step_ab.log
