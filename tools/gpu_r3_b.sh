#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python tools/gemm_timeline.py > gpurun_out/gemm_timeline.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_timeline.log | grep -A4 "qkv.*epi [36]"
