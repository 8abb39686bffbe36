#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 300 python tools/bench_ln.py 2>&1 | grep -v amdgpu | tee gpurun_out/ln.log
