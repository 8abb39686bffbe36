#!/bin/bash
# mid-round checkpoint: the whole GPU suite, then the config 2 / config 5 benches
mkdir -p gpurun_out; cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/mid_tests.log
timeout 400 python bench.py --steps 50 --warmup 5 > gpurun_out/mid_bench2.json 2> gpurun_out/mid_bench2.err
timeout 400 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/mid_bench5.json 2> gpurun_out/mid_bench5.err
timeout 300 python tools/bench_decode_step.py 4 2>&1 | grep -v "amdgpu.ids\|fused=False\|dependent" > gpurun_out/mid_decode.log
cat gpurun_out/mid_tests.log; tail -2 gpurun_out/mid_bench2.err
python - <<'PY'
import json
for f in ('gpurun_out/mid_bench2.json','gpurun_out/mid_bench5.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], d.get('exact_index_step',{}).get('ms_per_step'), d.get('roofline',{}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/mid_decode.log
