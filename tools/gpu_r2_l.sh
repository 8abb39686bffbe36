#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m mmvid_amd.build > gpurun_out/build.log 2>&1
F='amdgpu.ids|Warning|warn|run_backward'
echo "== attention tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "attention" 2>&1 | tail -3 | cut -c1-300
for s in 42 49 46 44 45 47 48 50 51 52; do echo "== graph seed $s"; timeout 300 python tools/stress_nan2.py 50 $s graph 2>&1 | grep -v -E "$F" | cut -c1-400; done
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 3 > gpurun_out/bench_l.log 2> gpurun_out/bench_l.err; echo "rc=$?"; grep "bench\]" gpurun_out/bench_l.err | cut -c1-220; python -c "
import json; d=json.loads(open('gpurun_out/bench_l.log').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], 'loss', d['loss']); [print('    ',k['kernel'],round(k['ms_per_step'],3),'ms',round(k['tflops'],1),'TF') for k in d['kernels']]"
for sc in 1 2; do echo "== bench strip sched $sc"; MMVID_STRIP_SCHED=$sc timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 3 2>&1 >/dev/null | grep "bench\]" | cut -c1-120; done
