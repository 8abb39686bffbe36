#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
echo "== bench default (fresh box, as the evidence run)"; timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_k.log 2> gpurun_out/bench_k.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_k.log').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], 'loss', d['loss'])"
for s in 42 43 44; do echo "== stress seed $s"; timeout 300 python tools/stress_nan.py 80 $s 10 2>&1 | grep -v -E "amdgpu.ids|Warning|warn" | cut -c1-600; done
echo "== stress seed 42, every 3"; timeout 300 python tools/stress_nan.py 60 42 3 2>&1 | grep -v -E "amdgpu.ids|Warning|warn" | cut -c1-600
