#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== framework launches per step"; timeout 600 python tools/step_ops.py 2>&1 | grep -v -E "amdgpu.ids|Warning|warn" | cut -c1-230 | head -70
