#!/bin/bash
# Round-6 session C: decode tests, decode step at batch 16 (graph replay), config-5 bench lines, rocprofv3 per-kernel durations
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$(pwd); o=gpurun_out; mkdir -p $o
timeout 600 python -m pytest tests/test_round6_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py tests/test_round4_gpu.py tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemv or decode or artv or kv_cache or attention_decode" 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -5
timeout 300 python tools/bench_decode_step.py 16 2>&1 | grep -v -i "warn\|amdgpu.ids" | grep -A1 "fused=True\|generate_images" | tee $o/decode_step_b16.log
for b in 4 16; do timeout 600 python bench.py --config 5 --batch $b --steps 2 --warmup 1 2>$o/bench_c5_b$b.err | tail -1 > $o/bench_c5_b$b.json; grep "bench\]" $o/bench_c5_b$b.err | cut -c1-200; done
rm -rf $o/prof_dec
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$o/prof_dec -o dec -- python $ROOT/bench.py --config 5 --batch 16 --steps 1 --warmup 1 > $ROOT/$o/prof_dec.log 2>&1; echo "rocprof rc=$?")
find $o/prof_dec -type f ! -name "*kernel_stats*" -delete
f=$(find $o/prof_dec -name "*kernel_stats*" | head -1); head -8 $f | cut -c1-200
