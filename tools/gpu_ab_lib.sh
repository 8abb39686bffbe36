#!/bin/bash
# Same-box A/B of two library builds: tools/gpu_ab_lib.sh <tag> "<command>" [reps]   (alternating runs, default library first)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; cmd=$2; reps=${3:-2}
for i in $(seq $reps); do
  echo "== default build, run $i"; eval "$cmd" 2>&1 | grep -v -i "warn\|amdgpu.ids"
  echo "== variant $tag, run $i"; MMVID_LIB=$PWD/mmvid_amd/libmmvid_hip.so.$tag eval "$cmd" 2>&1 | grep -v -i "warn\|amdgpu.ids"
done
