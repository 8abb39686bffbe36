#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu | tee gpurun_out/attn.log
timeout 600 python -m pytest tests -q -m gpu -x -k "tower or bert_training or config2" 2>&1 | tail -3
