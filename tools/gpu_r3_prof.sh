#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $ROOT/gpurun_out/prof/bench.json 2> $ROOT/gpurun_out/prof/bench.err)
find gpurun_out/prof -name "*kernel_stats.csv" | head; find gpurun_out/prof -name "*kernel_trace.csv" -delete; find gpurun_out/prof -name "*agent_info.csv" -delete
tail -2 gpurun_out/prof/bench.err
