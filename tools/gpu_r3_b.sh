#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 600 python tools/bench_gemm_epi.py > gpurun_out/gemm_epi.log 2>&1; echo "epi rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_epi.log | tail -16
