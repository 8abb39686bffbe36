#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "captured or gemv or artv or config5" -s > gpurun_out/sel.log 2>&1; echo "rc=$?"; grep -E "captured exchange|passed|failed|^FAILED|fused decode|config 5" gpurun_out/sel.log | cut -c1-250
echo "== decode step bench"; timeout 600 python tools/bench_decode_step.py 4 > gpurun_out/decode_step.log 2>&1; cut -c1-200 gpurun_out/decode_step.log | tail -16
timeout 600 python tools/bench_decode_step.py 1 > gpurun_out/decode_step_b1.log 2>&1; cut -c1-200 gpurun_out/decode_step_b1.log | tail -16
