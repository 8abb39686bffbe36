#!/bin/bash
# whole-step in-process A/B of library options: tools/gpu_r3_ab.sh "opt v1 v2" "opt v1 v2" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
: > gpurun_out/ab.log
for spec in "$@"; do
  timeout 400 python tools/ab_graph.py $spec 2>&1 | grep -E "median" | tee -a gpurun_out/ab.log
done
