#!/bin/bash
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "dpp or decode or artv or gemv or kv or sample_cached or config5 or long_cache" 2>&1 | tail -3 > gpurun_out/decode_tests.log
timeout 300 python tools/decode_gemv_timeline.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/decode_bench.log
timeout 300 python tools/bench_decode_step.py 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/decode_bench.log
timeout 300 python tools/bench_decode_step.py 1 2>&1 | grep -v amdgpu.ids >> gpurun_out/decode_bench.log
cat gpurun_out/decode_tests.log gpurun_out/decode_bench.log
