#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
for occ in 0 1 2 4 7; do echo "attn_occ=$occ: $(MMVID_ATTN_OCC=$occ timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu | head -1)"; done | tee gpurun_out/attn.log
