#!/bin/bash
# One parameterised GPU check (replaces the per-experiment gpu_r3_*.sh scripts):
#   tools/gpu_check.sh "<pytest -k expression>" ["<command>" ...]
# runs the selected `-m gpu` tests, then every further argument as a shell command; everything is logged under gpurun_out/check.log.
#   e.g.  gpurun -- 'bash tools/gpu_check.sh "attention or tower" "python tools/microbench.py attention" "python tools/ab_graph.py py:vae.stream f32 bf16"'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  echo "== pytest -m gpu -k \"$1\""
  timeout 1500 python -m pytest tests -x -q -m gpu -k "$1" 2>&1 | tail -4
  shift
  for c in "$@"; do
    echo "== $c"
    timeout 900 bash -c "$c" 2>&1 | grep -v amdgpu.ids | tail -40
  done
} 2>&1 | tee gpurun_out/check.log
