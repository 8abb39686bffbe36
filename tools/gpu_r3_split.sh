#!/bin/bash
# vae.strict = 'split': operator tests, golden index tests, and the training-step bench in that mode
mkdir -p gpurun_out; cd /root/repo
timeout 900 python -m pytest tests/test_round3_gpu.py -x -q -s -k "split" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/split_tests.log
timeout 400 python bench.py --strict split --steps 30 --warmup 5 > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err
tail -3 gpurun_out/bench_split.err
cat gpurun_out/split_tests.log
python -c "
import json
d=json.loads(open('gpurun_out/bench_split.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['config']['workload'])
"
