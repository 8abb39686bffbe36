#!/bin/bash
# Round-6 session D: full GPU suite + config-5 lines
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -15
for b in; do timeout 600 python bench.py --config 5 --batch $b --steps 2 --warmup 1 2>$o/bench_c5_b$b.err | tail -1 > $o/bench_c5_b$b.json; python -c "
import json;d=json.loads(open('$o/bench_c5_b$b.json').read().strip().splitlines()[-1]);print('batch $b', round(d['value']), 'tok/s', round(d['ms_per_step'],1),'ms', d['roofline']['frac'], d['roofline']['ms_per_token_step'])"; done
