#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
F='amdgpu.ids|Warning|warn|run_backward'
(for s in 101 102 103; do timeout 600 python tools/stress_nan2.py 400 $s graph 2 2>&1 | grep -v -E "$F" | cut -c1-300; done
for s in 201 202; do timeout 600 python tools/stress_nan2.py 300 $s graph 4 2>&1 | grep -v -E "$F" | cut -c1-300; done) | tee gpurun_out/stress_long.log
